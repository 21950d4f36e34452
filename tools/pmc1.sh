cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/pmc_c5 -- python3 $R/tools/one_layer.py --kind conv --k 5 --s 2 --cin 192 --cout 192 --n 36 --hw 256 384 --reps 4 > $R/gpurun_out/pmc_c5.txt 2>&1
python3 $R/tools/one_layer.py --kind conv --k 5 --s 2 --cin 192 --cout 192 --n 36 --hw 256 384 --reps 6 >> $R/gpurun_out/pmc_c5.txt 2>&1
python3 $R/tools/profile_layers.py > $R/gpurun_out/r02_layers_c.txt 2>&1
echo done
