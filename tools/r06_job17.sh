#!/bin/bash
# ln_reduce_kernel with eight loads really in flight: training parity tests, kernel stats at 1024 and 8192 samples (GPU box)
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "train" > $O/lnr_tests.txt 2>&1
tail -3 $O/lnr_tests.txt
unset BESO_HIP_LIB
timeout 200 bash tools/r05_train_stats.sh lnr_1024 1024 kitchen > $O/lnr_1024.txt 2>&1; tail -1 $O/lnr_1024_line.json; grep -E "ln_reduce|slab_reduce" $O/lnr_1024_kernel_stats.txt
timeout 300 bash tools/r05_train_stats.sh lnr_8192 8192 kitchen > $O/lnr_8192.txt 2>&1; tail -1 $O/lnr_8192_line.json; grep -E "ln_reduce|slab_reduce" $O/lnr_8192_kernel_stats.txt
