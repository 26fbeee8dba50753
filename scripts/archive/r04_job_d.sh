#!/bin/bash
mkdir -p gpurun_out/r04
timeout -k 10 300 python -m pytest tests/test_gpu_drivers.py tests/test_gpu_ring_ranks.py -x -q -m gpu > gpurun_out/r04/drivers2.log 2>&1; tail -3 gpurun_out/r04/drivers2.log
for p in 0 1; do ./lattice-boltzmann-method_amd/drivers/bin/slab_ring_rt --emulate 4 --rows 2048 --cols 2048 --steps 30 --warmup 5 --check 1 --parts $p > gpurun_out/r04/rt_emulate4_padded_parts$p.json 2>&1; cat gpurun_out/r04/rt_emulate4_padded_parts$p.json; done
LBM_TUNE=row_pad=0 ./lattice-boltzmann-method_amd/drivers/bin/slab_ring_rt --emulate 4 --rows 2048 --cols 2048 --steps 30 --warmup 5 --parts 1 > gpurun_out/r04/rt_emulate4_dense_parts1.json 2>&1; cat gpurun_out/r04/rt_emulate4_dense_parts1.json
LBM_HIP_LIB=$PWD/lattice-boltzmann-method_amd/lib_exp/liblbm_hip.so timeout -k 10 600 python -m pytest tests/test_gpu_cg.py tests/test_gpu_bgk.py tests/test_gpu_ibm.py tests/test_gpu_kbc.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/r04/exp_build_tests2.log 2>&1; tail -3 gpurun_out/r04/exp_build_tests2.log
timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --force-halo --no-pmc --no-secondary --no-cpu-baseline > gpurun_out/r04/bench_self_ring.json 2>gpurun_out/r04/bench_self_ring.err; python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/r04/bench_self_ring.json') if l.startswith('{')][-1]); print('self ring', d['value'], d['ms_per_step'], d['config']['transport'][:60], d.get('ring_phases'))"
timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-pmc --no-secondary --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no ring', d['value'], d['ms_per_step'])"
