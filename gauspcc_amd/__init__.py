"""gauspcc_amd -- the GausPcgc hot path of Wangkkklll/GausPcc on MI355X.

Importing the package exports GPU_MAX_HW_QUEUES=8 unless the caller has set it: the HIP runtime maps a process's streams
onto four hardware queues by default, a context of this library uses three streams, and two contexts that share queues run
in lockstep (19.2 against 22.1 Mpoints/s).  The variable is read when HIP initialises, so it only helps
a host application that imports this package before its first GPU call -- one that cannot should export it itself."""
import os as _os

_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
