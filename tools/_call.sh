cd $GRAFT_REPO_ROOT
python tools/time_probe.py C3 40 0
python tools/time_probe.py C3 30 20
python tools/time_probe.py C3 20 200
