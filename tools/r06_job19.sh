#!/bin/bash
# Extended fuzz of the final binary (GPU box, repo root): ~25 minutes
O=gpurun_out; mkdir -p $O
( timeout 700 python tools/fuzz_small.py 480 17; timeout 800 python tools/fuzz_train_bf16.py 600 23; PYTORCH_NO_CUDA_MEMORY_CACHING=1 timeout 500 python tests/fuzz_shapes.py 300 29; timeout 400 python tools/fuzz_train.py 240 31; timeout 500 python tests/fuzz_forward.py 400 37; timeout 300 python tests/determinism.py --reps 16 ) 2>&1 | grep -v amdgpu.ids | grep -i "fuzz\|worst\|determin\|identical\|FAIL\|Error\|Traceback" > $O/r06_fuzz_long.txt
cat $O/r06_fuzz_long.txt
