"""Host-side containers and the device-resident prioritized buffer.

``RandomAccessQueue``  FIFO with O(1) indexing for the CPU (gpu=-1) path
``TreeFrame``          integer bookkeeping of the reference's sliding tree frame
``PrioritizedBuffer``  sum / min trees in HBM (imported lazily: needs a GPU)
``HostPrioritizedBuffer`` (``host_prioritized``)  the same interface on the host, for buffers
                       that are used without a GPU (gpu=None plumbing path)
"""
from pfrl_amd.collections.random_access_queue import RandomAccessQueue  # NOQA
from pfrl_amd.collections.tree_frame import TreeFrame  # NOQA
from pfrl_amd.collections.persistent_collections import PersistentRandomAccessQueue  # NOQA


def __getattr__(name):
    # lazily: the module pulls in the HIP bindings
    if name == "prioritized":
        import importlib

        return importlib.import_module("pfrl_amd.collections.prioritized")
    if name == "PrioritizedBuffer":
        from pfrl_amd.collections.prioritized import PrioritizedBuffer

        return PrioritizedBuffer
    raise AttributeError(name)
