#!/bin/bash
# Store-wave instance of the training forward: parity tests, then kernel stats against the SW = 0 variant (GPU box, repo root)
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "training_forward_as_one_launch or training_step_kitchen_1024 or hip_training_step_matches_reference or training_step_random_shapes or bf16_training_step_on_random" > $O/sw_tests.txt 2>&1
tail -5 $O/sw_tests.txt
bash tools/r06_job15.sh base sw0
