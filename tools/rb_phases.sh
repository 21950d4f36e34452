# DIAG build on the GPU box into a library of its own (the shipped library has no stamps and is left alone), then tools/rb_phases.py
cd $GRAFT_REPO_ROOT/shallow-ntc_amd/csrc && make DIAG=1 LIB=../lib/libsntc_hip_diag.so BUILD=build_diag -j16 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT && SNTC_LIB=$GRAFT_REPO_ROOT/shallow-ntc_amd/lib/libsntc_hip_diag.so python tools/rb_phases.py 2>&1 | tail -2
