"""MI355X-native REFIL learner hot path (plugin surface of the reference over librefil_hip.so)."""
import os as _os

# HIP maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4 = the learner step's four streams). A
# torch.distributed process group on RCCL creates streams of its own; with four queues they share queues with the step's
# streams and every step is 17-40 % slower, collective or not (tools/probes/dp_overhead.py). Eight queues: unchanged
# single-process, ~10 us per all-reduce. The variable is read when HIP initialises -- import refil_amd (or set it) before
# the first CUDA call of the process.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
