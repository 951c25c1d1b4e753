"""mmhip: MI355X-native late-fusion fine-tuning path."""
# (Round 4 briefly set GPU_MAX_HW_QUEUES=8 here, as a second guard against streams sharing one of ROCm's 4 default hardware queues.  Withdrawn:
# with two processes on ONE GPU -- tools/bench_two_ranks_one_gpu.sh, gloo moving device tensors -- the step deadlocked with 8 queues per process
# and runs with the default 4.  The stream pool of csrc/mmhip_common.h is what keeps a process's queue mapping stable; the variable is the user's.)
