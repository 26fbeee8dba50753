# walking-tile kernel of the two-phase step (cg_strip2 = 41 / 42) against the tile kernel: MLUPS + FETCH/WRITE per step
mkdir -p gpurun_out/probe
timeout -k 10 300 python -m pytest tests/test_gpu_cg.py -x -q -k "strip_kernel_equals" 2>&1 | tail -3 || exit 1
for cfg in ${CFGS:-0:0 41:0 42:0 41:256 42:256 42:512 42:1024}; do set -- ${cfg/:/ }
  timeout -k 10 200 python bench.py --secondary-only --secondary cg --tune cg_strip2=$1 --tune cg_rows2=$2 ${EXTRA:-} 2>&1 | python -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        s=json.loads(ln)['secondary'][0]; r=s['roofline']; p=r.get('pmc') or {}
        print('cg_strip2=$1 rows=$2', s['value'], 'MLUPS; fetch GB', round(p.get('fetch_bytes',0)/1e9,3), 'write GB', round(p.get('write_bytes',0)/1e9,3), 'achieved', r.get('achieved'))
" || exit 1
done
