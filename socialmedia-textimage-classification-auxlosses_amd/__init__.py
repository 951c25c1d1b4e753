"""mmhip: MI355X-native late-fusion fine-tuning path."""
import os as _os

# The engines keep up to four streams busy (the caller's + three pooled side streams, csrc/mmhip_common.h: pool_stream); the input
# pipeline adds a copy stream and data parallelism RCCL's.  ROCm multiplexes HIP streams onto 4 hardware queues by default, and two
# streams that land on one queue run in order -- the overlap is lost without any error (measured: strict-dtype step 29.4 vs 25.0 ms,
# profiles/r04_hw_queues.txt).  Ask for 8 unless the user has chosen; the variable is read when the HIP runtime initialises, so this
# only takes effect when the package is imported before the first GPU call (it is a no-op otherwise).
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
