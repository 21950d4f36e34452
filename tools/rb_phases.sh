# DIAG build on the GPU box (the shipped library has no stamps), then tools/rb_phases.py
cd $GRAFT_REPO_ROOT/shallow-ntc_amd/csrc && make clean > /dev/null && make DIAG=1 -j16 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT && python tools/rb_phases.py 2>&1 | tail -2
