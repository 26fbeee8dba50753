#!/bin/bash
# round 4, third evidence job: PMC per XCD order with the solver's rows padded; the supervised bench under torch.distributed.run
# with REAL workers (2 ranks sharing GPU 0), the driver's launch shape
mkdir -p gpurun_out/r04
scripts/r04_cg_order_pmc.sh > /dev/null; cp gpurun_out/r04/cg_order_pmc.txt gpurun_out/r04/cg_order_pmc_pitched.txt; cat gpurun_out/r04/cg_order_pmc_pitched.txt
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 20 --warmup 5 --share-gpu --rows 2048 --cols 4096 > gpurun_out/r04/bench_torchrun_share2.json 2> gpurun_out/r04/bench_torchrun_share2.err; echo "torchrun rc=$?"; grep '^{' gpurun_out/r04/bench_torchrun_share2.json | tail -1 | cut -c1-300; grep -o '"launcher": {[^}]*}' gpurun_out/r04/bench_torchrun_share2.json | tail -1
