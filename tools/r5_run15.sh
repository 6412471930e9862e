cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/notorch_probe.py 2>&1 | tail -2
python tools/path_probe.py vga 45 2>&1 | tail -1
python tools/notorch_probe.py hd 2>&1 | tail -1
python tools/path_probe.py hd 15 2>&1 | tail -1
python tools/host_paths.py 2>&1 | tail -1 | cut -c1-400
