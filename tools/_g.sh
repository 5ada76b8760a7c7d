cd /root/repo
for v in "" tools/_variants/noface.so tools/_variants/nointerior.so; do
echo "== $v"
CONCEPT_GPU_LIB=$v python tools/sr_rung_cost.py uniform 2>&1 | grep -v 'plain list' | tail -5 | cut -c1-110
done
