# same-box A/B of the HIP runtime under bench.py at N = 1: no PyTorch in the process vs --with-torch
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
L="--cpu-frames 0 --host-frames 0 --no-secondary --streams 0"
for i in 1 2 3; do
python bench.py $L 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('system runtime ', d['value'], d['value_min_max'], d['roofline']['avg_launch_us'], d['host_enqueue_us_per_frame'])"
python bench.py $L --with-torch 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('torch runtime  ', d['value'], d['value_min_max'], d['roofline']['avg_launch_us'], d['host_enqueue_us_per_frame'])"
done
python tools/notorch_probe.py | tail -1
python tools/path_probe.py vga 45 2>&1 | tail -1
python bench.py $L --config hd2mm 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('hd system runtime ', d['value'], d['roofline']['avg_launch_us'])"
python bench.py $L --config hd2mm --with-torch 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('hd torch runtime  ', d['value'], d['roofline']['avg_launch_us'])"
