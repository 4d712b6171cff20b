one() { python bench.py --no-cpu --no-extra --steps 3000 --warmup 300 "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; c=j['config']; print('kernel_us=%6.3f frac=%.3f tune=%s' % (r['kernel_us'], r['frac'], c['tune']))
"; }
one; for p in 2 3 4 5 7 9 -2 -4 -6; do one --tune prio=$p; done; one
