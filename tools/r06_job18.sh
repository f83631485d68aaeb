#!/bin/bash
# Kept h / GELU(h) of the three-tile forward as 16-byte pieces for the tile pair: parity, kernel time and step A/B (GPU box)
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "training_forward_as_one_launch or training_step_kitchen_1024 or hip_training_step_matches_reference or training_step_random_shapes or bf16_training_step_on_random" > $O/hyb_tests.txt 2>&1
tail -2 $O/hyb_tests.txt
bash tools/r06_job15.sh base hyb0
bash tools/r05_train_ab.sh > $O/hyb_ab.txt 2>&1; cat $O/hyb_ab.txt
