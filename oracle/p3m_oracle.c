/* placeholder, filled in with the P3M short-range restatement */
#include <stdint.h>
int64_t orc_p3m_placeholder(void) { return 0; }
