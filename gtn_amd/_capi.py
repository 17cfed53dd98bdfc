"""ctypes binding of the C ABI declared in include/gtn_amd.h.

`load(path)` returns a ctypes library object with every entry point of the
header typed.  The product library is gtn_amd/lib/libgtn_amd.so (HIP, gfx950);
the parity tests also load oracle/_ref/libgtn_ref.so -- the unmodified
reference behind the same ABI -- through this same function.
"""
import ctypes as C
import os

c_graph = C.c_void_p
c_graph_p = C.POINTER(C.c_void_p)
c_i32_p = C.POINTER(C.c_int)
c_i64_p = C.POINTER(C.c_int64)
c_u8_p = C.POINTER(C.c_uint8)
c_f32_p = C.POINTER(C.c_float)

GRAD_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, c_graph_p, C.c_int, c_graph)
CTX_FREE = C.CFUNCTYPE(None, C.c_void_p)

OK, INVALID_ARGUMENT, LOGIC_ERROR, RUNTIME_ERROR, OUT_OF_RANGE, DEVICE_ERROR = range(6)

# name -> argtypes (restype is c_int status unless listed in _RESTYPE)
_SIGS = {
    "gtnx_set_device": [C.c_int],
    "gtnx_get_device": [c_i32_p],
    "gtnx_set_stream": [C.c_void_p],
    "gtnx_compose_mode": [C.c_int, C.POINTER(C.c_int)],
    "gtnx_synchronize": [],
    "gtnx_memory_stats": [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)],
    "gtnx_empty_cache": [],
    "gtnx_reclaim": [],
    "gtnx_graph_create": [C.c_int, c_graph_p],
    "gtnx_graph_copy": [c_graph, c_graph_p],
    "gtnx_graph_deep_copy": [c_graph, c_graph_p],
    "gtnx_graph_destroy": [c_graph],
    "gtnx_graph_add_node": [c_graph, C.c_int, C.c_int, c_i32_p],
    "gtnx_graph_add_arc": [c_graph, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, c_i32_p],
    "gtnx_graph_add_nodes": [c_graph, C.c_int, C.c_void_p, C.c_void_p],
    "gtnx_graph_add_arcs": [c_graph, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    "gtnx_graph_num_nodes": [c_graph, c_i64_p],
    "gtnx_graph_num_arcs": [c_graph, c_i64_p],
    "gtnx_graph_num_start": [c_graph, c_i64_p],
    "gtnx_graph_num_accept": [c_graph, c_i64_p],
    "gtnx_graph_item": [c_graph, c_f32_p],
    "gtnx_graph_arc_sort": [c_graph, C.c_int],
    "gtnx_graph_mark_arc_sorted": [c_graph, C.c_int],
    "gtnx_graph_ilabel_sorted": [c_graph, c_i32_p],
    "gtnx_graph_olabel_sorted": [c_graph, c_i32_p],
    "gtnx_graph_weights": [c_graph, C.c_int, C.POINTER(c_f32_p)],
    "gtnx_graph_get_weights": [c_graph, C.c_void_p],
    "gtnx_graph_set_weights": [c_graph, C.c_void_p],
    "gtnx_graph_set_weights_device": [c_graph, C.c_void_p],
    "gtnx_graph_weights_device": [c_graph, C.POINTER(C.c_void_p)],
    "gtnx_graph_labels_to_array": [c_graph, C.c_void_p, C.c_int],
    "gtnx_graph_get_start": [c_graph, C.c_void_p],
    "gtnx_graph_get_accept": [c_graph, C.c_void_p],
    "gtnx_graph_is_start": [c_graph, C.c_int, c_i32_p],
    "gtnx_graph_is_accept": [c_graph, C.c_int, c_i32_p],
    "gtnx_graph_make_accept": [c_graph, C.c_int],
    "gtnx_graph_num_out": [c_graph, C.c_int, c_i64_p],
    "gtnx_graph_num_in": [c_graph, C.c_int, c_i64_p],
    "gtnx_graph_get_out": [c_graph, C.c_int, C.c_void_p],
    "gtnx_graph_get_in": [c_graph, C.c_int, C.c_void_p],
    "gtnx_graph_get_arcs": [c_graph, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    "gtnx_graph_get_arc": [c_graph, C.c_int, c_i32_p, c_i32_p, c_i32_p, c_i32_p, c_f32_p],
    "gtnx_graph_set_weight": [c_graph, C.c_int, C.c_float],
    "gtnx_graph_calc_grad": [c_graph, c_i32_p],
    "gtnx_graph_set_calc_grad": [c_graph, C.c_int],
    "gtnx_graph_is_grad_available": [c_graph, c_i32_p],
    "gtnx_graph_grad": [c_graph, c_graph_p],
    "gtnx_graph_zero_grad": [c_graph],
    "gtnx_graph_add_grad": [c_graph, C.c_void_p, C.c_int64],
    "gtnx_graph_add_grad_graph": [c_graph, c_graph],
    "gtnx_graph_id": [c_graph, C.POINTER(C.c_size_t)],
    "gtnx_graph_create_op": [c_graph_p, C.c_int, GRAD_FN, C.c_void_p, CTX_FREE, c_graph_p],
    "gtnx_graph_num_inputs": [c_graph, c_i64_p],
    "gtnx_graph_get_input": [c_graph, C.c_int, c_graph_p],
    "gtnx_graph_set_inputs": [c_graph, c_graph_p, C.c_int],
    "gtnx_graph_set_grad_fn": [c_graph, GRAD_FN, C.c_void_p, CTX_FREE],
    "gtnx_graph_has_grad_fn": [c_graph, c_i32_p],
    "gtnx_scalar_graph": [C.c_float, C.c_int, c_graph_p],
    "gtnx_linear_graph": [C.c_int, C.c_int, C.c_int, c_graph_p],
    "gtnx_linear_graph_n": [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, c_graph_p],
    "gtnx_linear_graph_borrow_n": [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, c_graph_p],
    "gtnx_negate": [c_graph, c_graph_p],
    "gtnx_add": [c_graph, c_graph, c_graph_p],
    "gtnx_subtract": [c_graph, c_graph, c_graph_p],
    "gtnx_compose": [c_graph, c_graph, c_graph_p],
    "gtnx_intersect": [c_graph, c_graph, c_graph_p],
    "gtnx_forward_score": [c_graph, c_graph_p],
    "gtnx_viterbi_score": [c_graph, c_graph_p],
    "gtnx_viterbi_path": [c_graph, c_graph_p],
    "gtnx_negate_n": [c_graph_p, C.c_int, c_graph_p],
    "gtnx_add_n": [c_graph_p, C.c_int, c_graph_p, C.c_int, c_graph_p],
    "gtnx_subtract_n": [c_graph_p, C.c_int, c_graph_p, C.c_int, c_graph_p],
    "gtnx_compose_n": [c_graph_p, C.c_int, c_graph_p, C.c_int, c_graph_p],
    "gtnx_intersect_n": [c_graph_p, C.c_int, c_graph_p, C.c_int, c_graph_p],
    "gtnx_forward_score_n": [c_graph_p, C.c_int, c_graph_p],
    "gtnx_viterbi_score_n": [c_graph_p, C.c_int, c_graph_p],
    "gtnx_viterbi_path_n": [c_graph_p, C.c_int, c_graph_p],
    "gtnx_items_n": [c_graph_p, C.c_int, C.c_void_p],
    "gtnx_items_device_n": [c_graph_p, C.c_int, C.c_void_p],
    "gtnx_grads_device_n": [c_graph_p, C.c_int, C.c_void_p, C.c_void_p],
    "gtnx_grads_bind_device_n": [c_graph_p, C.c_int, C.c_void_p, C.c_void_p],
    "gtnx_clone": [c_graph, C.c_int, c_graph_p],
    "gtnx_concat": [c_graph_p, C.c_int, c_graph_p],
    "gtnx_closure": [c_graph, c_graph_p],
    "gtnx_union": [c_graph_p, C.c_int, c_graph_p],
    "gtnx_batch_from_graphs": [c_graph_p, C.c_int, c_graph_p],
    "gtnx_batch_ctc_targets": [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, c_graph_p],
    "gtnx_batch_asg_force_align": [C.c_void_p, C.c_void_p, C.c_int, c_graph, C.c_int, c_graph_p],
    "gtnx_batch_linear": [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, c_graph_p],
    "gtnx_batch_destroy": [c_graph],
    "gtnx_batch_size": [c_graph, c_i32_p],
    "gtnx_batch_get": [c_graph, C.c_int, c_graph_p],
    "gtnx_batch_negate": [c_graph, c_graph_p],
    "gtnx_batch_add": [c_graph, c_graph, c_graph_p],
    "gtnx_batch_subtract": [c_graph, c_graph, c_graph_p],
    "gtnx_batch_subtract_into": [c_graph, c_graph, C.c_void_p, c_graph_p],
    "gtnx_batch_compose": [c_graph, c_graph, c_graph_p],
    "gtnx_batch_intersect": [c_graph, c_graph, c_graph_p],
    "gtnx_batch_forward_score": [c_graph, c_graph_p],
    "gtnx_batch_viterbi_score": [c_graph, c_graph_p],
    "gtnx_batch_viterbi_path": [c_graph, c_graph_p],
    "gtnx_batch_backward": [c_graph, C.c_int],
    "gtnx_batch_items": [c_graph, C.c_void_p],
    "gtnx_batch_items_device": [c_graph, C.c_void_p],
    "gtnx_batch_grads_bind_device": [c_graph, C.c_void_p, C.c_void_p],
    "gtnx_batch_grads_device": [c_graph, C.c_void_p, C.c_void_p],
    "gtnx_backward": [c_graph, C.c_int],
    "gtnx_backward_with_grad": [c_graph, c_graph, C.c_int],
    "gtnx_backward_n": [c_graph_p, C.c_int, C.c_int],
    "gtnx_equal": [c_graph, c_graph, c_i32_p],
    "gtnx_isomorphic": [c_graph, c_graph, c_i32_p],
    "gtnx_remove": [c_graph, C.c_int, C.c_int, c_graph_p],
    "gtnx_graph_load_buffer": [C.c_void_p, C.c_size_t, c_graph_p],
    "gtnx_parallel_enter": [],
    "gtnx_parallel_leave": [],
    "gtnx_parallel_flush": [],
    "gtnx_prof_enable": [C.c_int],
    "gtnx_prof_reset": [],
    "gtnx_prof_get": [C.c_char_p, C.POINTER(C.c_double), c_i64_p, C.POINTER(C.c_double)],
    "gtnx_prof_names": [C.c_char_p, C.c_size_t],
    "gtnx_debug_symbolic_route": [c_graph, C.c_int, c_i32_p],
    "gtnx_debug_route_name": [C.c_int, C.c_char_p, C.c_size_t],
    "gtnx_debug_viterbi_ties": [c_i64_p, c_i64_p],
    "gtnx_debug_tie_ranks": [c_graph, C.c_void_p, C.c_void_p, C.c_void_p],
    "gtnx_comm_create": [c_i32_p, C.c_int, C.POINTER(C.c_void_p)],
    "gtnx_comm_destroy": [C.c_void_p],
    "gtnx_comm_size": [C.c_void_p, c_i32_p],
    "gtnx_comm_all_gather_f32": [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int64],
    "gtnx_comm_all_reduce_sum_f32": [C.c_void_p, C.POINTER(C.c_void_p), C.c_int64],
}

_RESTYPE = {
    "gtnx_last_error": (C.c_char_p, []),
    "gtnx_version": (C.c_char_p, []),
    "gtnx_backend": (C.c_char_p, []),
    "gtnx_device_count": (C.c_int, []),
}

ALL_SYMBOLS = sorted(list(_SIGS) + list(_RESTYPE))


def default_library_path():
    # the HIP engine, built in-tree; nothing else (the reference-backed test shim is bound by
    # tests/refbackend/gtn_ref.py, test infrastructure, by passing its path to load())
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libgtn_amd.so")


def load(path=None):
    path = path or default_library_path()
    if not os.path.exists(path):
        raise ImportError(
            f"gtn_amd: native library not found at {path}. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950). "
            "There is no CPU fallback."
        )
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL if hasattr(C, "RTLD_GLOBAL") else 0)
    for name, args in _SIGS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            # engine extensions (batch records, borrowed tensors, collectives between devices): absent from the reference-backed
            # test shim (oracle/ref_shim.cpp), which only tests/refbackend/gtn_ref.py loads
            if name.startswith("gtnx_batch_") or name.startswith("gtnx_comm_") or name in ("gtnx_linear_graph_borrow_n", "gtnx_grads_bind_device_n",
                                                             "gtnx_parallel_enter", "gtnx_parallel_leave", "gtnx_parallel_flush"):
                continue
            raise
        fn.argtypes = args
        fn.restype = C.c_int
    for name, (res, args) in _RESTYPE.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = res
    lib._path = path
    return lib
