one() { python bench.py --workload c4 --no-cpu --no-extra --steps 100 --warmup 20 "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; c=j['config']; print('$TAG %-36s kernel_us=%7.2f frac=%.3f seg_rows=%d' % (r['kernel'], r['kernel_us'], r['frac'], c['tile_rows']))
"; }
for rep in 1 2; do
TAG=ring4 one
TAG=ring6 VARANNEAL_AMD_LIB=$PWD/varanneal_amd/libvaranneal_amd_ring6.so one
TAG=ring6-seg250 VARANNEAL_AMD_LIB=$PWD/varanneal_amd/libvaranneal_amd_ring6.so one --tile-rows 250
done
