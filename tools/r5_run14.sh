cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2; do
python bench.py --cpu-frames 0 --no-secondary --streams 0 --steps 5 --reps 2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('lean bench: pinned', d['pinned_h2d_path']['h2d_gbps'], 'host', d['host_image_path']['frames_per_s'])"
python tools/pinned_probe.py 2048 | tail -1
done
