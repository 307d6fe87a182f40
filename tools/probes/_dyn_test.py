import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch
import test_gpu_ops as t
t.test_attention_persistent_walk(128, 80, 16, 8, 4, 16, True)
print("big walk ok")
