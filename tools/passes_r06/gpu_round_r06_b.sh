# r6 evidence pass B (one gpurun call): long fuzz runs (all scene kinds, then Mandelbulb only), emulated N-GPU balance, the N > 1 bench modes on the one GPU,
# the RCCL gather at world 1, host-buffer rates, the whole c5 frame, cold frames.
# usage: bash tools/passes_r06/gpu_round_r06_b.sh <tag>      outputs under gpurun_out/
set -x
TAG=${1:-v1}
R=r06
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python tools/fuzz_parity.py 2400 120000 2>&1 | tail -3 | tee gpurun_out/${R}_fuzz_2400_$TAG.txt
timeout 900 python tools/fuzz_parity.py 800 130000 bulb 2>&1 | tail -3 | tee gpurun_out/${R}_fuzz_bulb_800_$TAG.txt
(for A in "8 c4" "8 bulb3" "2 c3" "4 c3" "8 c3" "8 c2"; do timeout 900 python tools/share_balance.py $A 2>&1 | tail -1; done) | tee gpurun_out/${R}_share_balance_$TAG.txt
# the N>1 bench modes on the one GPU: self-launch (gloo, shared GPU), one process over two entries, the RCCL path at world 1 with the gather alone
timeout 600 python bench.py --gpus 2 --workload c2 --steps 2 --warmup 1 --backend gloo --share-gpu --check-film --cpu-seconds 0 --no-roofline 2>/dev/null | tail -1 | tee gpurun_out/${R}_bench_c2_2ranks_gloo_$TAG.json | cut -c1-1800
timeout 600 python bench.py --gpus 2 --single-process --share-gpu --workload c2 --steps 2 --warmup 1 --cpu-seconds 0 --check-film 2>/dev/null | tail -1 | tee gpurun_out/${R}_bench_c2_single_process_2_$TAG.json | cut -c1-1800
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 1 --workload c3 --backend nccl --force-dist --gather-only 2>/dev/null | tail -1 | tee gpurun_out/${R}_gather_only_world1_$TAG.json | cut -c1-1500
# the driver's N > 1 command line at world 1 on the box's RCCL (process group, barrier, all-reduce, FilmGather incl. rank 0's own block), bulb3 for a short run
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29578 bench.py --gpus 1 --workload bulb3 --steps 2 --warmup 1 --force-dist --cpu-seconds 0 2>/dev/null | tail -1 | tee gpurun_out/${R}_bench_bulb3_force_dist_world1_$TAG.json | cut -c1-600
timeout 600 python tools/host_rate.py c3 2>&1 | tail -1
timeout 600 python tools/host_rate.py c2 2>&1 | tail -1
timeout 900 python bench.py --workload c5 --steps 1 --warmup 0 --no-roofline --no-cold --cpu-seconds 0 2>&1 | tail -1 > gpurun_out/${R}_bench_c5_$TAG.json; cut -c1-220 gpurun_out/${R}_bench_c5_$TAG.json
for i in 1 2 3; do timeout 120 python tools/cold_breakdown.py shipped 0 2>&1 | tail -1; done
sleep 6
for WL in c3 c2 c2 shipped shipped; do timeout 300 python tools/cold_frame.py $WL -1 0 2>&1 | tail -1; done
