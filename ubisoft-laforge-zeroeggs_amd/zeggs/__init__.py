"""MI355X engine behind the reference's Python interface (modules, train, generate)."""
import os as _os

# The training engine spreads an iteration over three HIP streams (engine.TrainEngine) and RCCL brings its own; the HIP runtime
# multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and two of ours on one queue run one after
# the other.  Only a default: set before the runtime initialises, an explicit setting of the user wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
