"""gtn_amd -- MI355X-native WFST shortest-distance and composition engine with
the facebookresearch/gtn interface.

Importing this package loads gtn_amd/lib/libgtn_amd.so (hand-written HIP for
gfx950 behind the C ABI of include/gtn_amd.h) and binds the reference's Python
interface to it (see gtn_amd/api.py).  There is no CPU fallback: if the native
library is missing the import fails, and if no GPU is visible every device
operation raises.
"""
from . import _capi
from .api import make_api, load_txt as _load_txt

_lib = _capi.load()
_api = make_api(_lib)

Graph = _api.Graph
Batch = _api.Batch
epsilon = _api.epsilon
negate = _api.negate
add = _api.add
subtract = _api.subtract
subtract_into = getattr(_api, "subtract_into", None)  # (batches only: the values written into the caller's tensor)
compose = _api.compose
intersect = _api.intersect
forward_score = _api.forward_score
viterbi_score = _api.viterbi_score
viterbi_path = _api.viterbi_path
backward = _api.backward
scalar_graph = _api.scalar_graph
linear_graph = _api.linear_graph
linear_graph_n = _api.linear_graph_n
equal = _api.equal
isomorphic = _api.isomorphic
items = _api.items
items_to_device = _api.items_to_device
grads_to_device = _api.grads_to_device
parallel_for = _api.parallel_for
clone = _api.clone
project_input = _api.project_input
project_output = _api.project_output
closure = _api.closure
concat = _api.concat
union = _api.union
remove = _api.remove
sample = _api.sample
rand_equivalent = _api.rand_equivalent
load = _api.load
save = _api.save
loadtxt = _api.loadtxt
savetxt = _api.savetxt
write_dot = _api.write_dot
draw = _api.draw
backend = _api.backend
device_count = _api.device_count
synchronize = _api.synchronize
set_device = _api.set_device
set_stream = _api.set_stream
compose_mode = _api.compose_mode
memory_stats = _api.memory_stats
empty_cache = _api.empty_cache
prof_enable = _api.prof_enable
prof_reset = _api.prof_reset
prof_get = _api.prof_get
prof_names = _api.prof_names
debug_symbolic_route = _api.debug_symbolic_route
debug_viterbi_ties = _api.debug_viterbi_ties
debug_tie_ranks = _api.debug_tie_ranks


def load_txt(text):
    return _load_txt(_api, text)


__version__ = _lib.gtnx_version().decode()


def _shutdown():
    # orderly teardown before the HIP runtime's own exit handlers run: finish
    # queued work and hand pooled device / pinned memory back
    try:
        _lib.gtnx_set_stream(None)
        _lib.gtnx_synchronize()
        _lib.gtnx_empty_cache()
    except Exception:
        pass


import atexit as _atexit  # noqa: E402

_atexit.register(_shutdown)
