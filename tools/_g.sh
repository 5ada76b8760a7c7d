cd /root/repo
python - <<'PY' 2>&1 | tail -45
import cProfile, pstats, sys, io, runpy
sys.argv = ['tools/soak_p3m.py', '0.04']
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path('tools/soak_p3m.py', run_name='__main__')
finally:
    pr.disable()
s = io.StringIO()
ps = pstats.Stats(pr, stream=s).sort_stats('tottime')
ps.print_stats(28)
print(s.getvalue()[-5500:])
PY
