"""MI355X-native VACNIC training step."""
import os as _os

# The step runs on four HIP streams (compute, weight gradients, frozen towers, small-token branches); the data-parallel reducer's
# RCCL communicator brings internal streams of its own.  The HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues
# (default 4) and shares queues beyond that: with the default, a ONE-rank communicator — no bytes on any link — costs the step
# 4.3 ms (two of the step's streams end up sharing a queue); with 6 or 8 queues the same run takes 91 / 78 ms (more than five
# active hardware queues time-slice); with 5 it costs nothing (64.6 vs 64.4-65.1 ms without any reducer), and the single-GPU step
# is indifferent to the setting (profiles/r4_ddp_one_rank_rccl.txt).  Read by the runtime at its first HIP call, so it is set
# when the package is imported; an explicit setting in the environment wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "5")
