cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
# does RCCL take two ranks on ONE device?  (a true RCCL rehearsal of the N = 2 paths if it does)
RATSDF_BENCH_DEVICE=0 RATSDF_BENCH_BACKEND=nccl NCCL_DEBUG=WARN timeout -k 10 150 python bench.py --gpus 2 --shard --steps 2 --warmup 1 --reps 1 --frames-per-step 12 --bcast-chunk 4 --cpu-frames 4 > gpurun_out/r5_nccl_one_device.log 2>&1; echo "rc=$?"; tail -c 1500 gpurun_out/r5_nccl_one_device.log
