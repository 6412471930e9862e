cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
SECONDS=0
python bench.py > gpurun_out/r5_bench_final.json 2> gpurun_out/r5_bench_final.err; echo "bench rc=$? seconds=$SECONDS"
