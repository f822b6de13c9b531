"""gauspcc_amd -- the GausPcgc hot path of Wangkkklll/GausPcc on MI355X.

Importing the package has no side effects on the host application.  A process that keeps MORE THAN ONE codec context busy
on a GPU (two scenes in flight, `--jobs 2`) should call `gauspcc_amd.export_hw_queues()` -- or export GPU_MAX_HW_QUEUES=8
itself -- BEFORE its first GPU call: the HIP runtime maps a process's streams onto four hardware queues by default, a
context of this library uses three streams, and two contexts that share queues run in lockstep (19.2 against 22.1
Mpoints/s, HISTORY.md section 7).  The command-line tools and bench.py do this for themselves."""
import os as _os


def export_hw_queues(contexts=2):
    """Opt-in: export GPU_MAX_HW_QUEUES (at least 8, four per context) unless the caller's environment already sets it.
    The variable is read when HIP initialises: call this before anything touches the GPU.  Returns the value in effect."""
    _os.environ.setdefault("GPU_MAX_HW_QUEUES", str(max(8, 4 * int(contexts))))
    return _os.environ["GPU_MAX_HW_QUEUES"]
