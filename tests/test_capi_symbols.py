"""CPU: the C-ABI library loads and exports every symbol include/nlt_hip.h declares, and the
Python binding table covers exactly that set (no compute calls without a GPU)."""
import ctypes
import os
import re

import nlt_amd
from nlt_amd import capi as C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    hdr = open(os.path.join(ROOT, 'include', 'nlt_hip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    return set(re.findall(r'\b(nlt_[a-z0-9_]+)\s*\(', hdr))


def test_library_is_built_in_tree():
    assert os.path.exists(C.LIB_PATH), "run `python __graft_entry__.py` (build()) first"
    assert os.path.dirname(C.LIB_PATH) == os.path.join(ROOT, 'neural-light-transport_amd')


def test_every_header_symbol_is_exported_and_bound():
    syms = header_symbols()
    assert len(syms) >= 11
    L = ctypes.CDLL(C.LIB_PATH)
    for s in syms:
        assert hasattr(L, s), "libnlt_hip.so does not export %s" % s
    assert syms == set(C.SIGNATURES), (syms ^ set(C.SIGNATURES))


def test_the_documents_name_only_entry_points_that_exist():
    """INTEGRATION.md (the reference-side binding guide), README.md and DESIGN.md must not send a maintainer to an entry point the
    header no longer declares (r06 removed several with their experiments)."""
    syms = header_symbols()
    not_functions = {'nlt_hip', 'nlt_status', 'nlt_amd', 'nlt_test', 'nlt_test_infer', 'nlt_repack_desc', 'nlt_tape_call', 'nlt_common', 'nlt_oracle',
                     'nlt_wino_fragment', 'nlt_mfma_fragment', 'nlt_tile_fragment'}     # (types, modules, device helpers of csrc/)
    removed_and_said_so = {'nlt_back_backward_parts', 'nlt_front5_forward', 'nlt_conv_forward_pair'}    # (named as history, with "removed")
    for doc in ('INTEGRATION.md', 'README.md', 'DESIGN.md'):
        text = open(os.path.join(ROOT, doc)).read()
        for name in set(re.findall(r'`(nlt_[a-z0-9_]+)`', text)) | set(re.findall(r'\b(nlt_[a-z0-9_]+)\(', text)):
            if name in not_functions or name in syms or name.endswith('_'):
                continue
            assert doc != 'INTEGRATION.md' and name in removed_and_said_so, "%s names %s, which include/nlt_hip.h does not declare" % (doc, name)


def test_metadata_calls_work_without_gpu():
    L = C.lib()
    assert b'gfx950' in L.nlt_version()
    assert L.nlt_status_string(0) == b'ok'
    assert L.nlt_status_string(-2).startswith(b'unsupported')
    # packed sizes: taps * ceil16(c0)+ceil16(c1) chunks * N/16 tiles * 256 floats
    assert C.packed_weight_floats(C.CONV_K2S2, 32, 0, 16) == 4 * 2 * 1 * 256
    assert C.packed_weight_floats(C.DECONV_K2S2, 8, 32, 4) == 1 * 3 * 1 * 256
    assert C.packed_weight_floats(C.CONV1X1, 4, 32, 12) == 1 * 3 * 1 * 256
    assert C.packed_weight_floats(9, 4, 0, 4) == -1


def test_no_cpu_fallback_for_cpu_tensors():
    import pytest
    import torch
    x = torch.zeros(4)
    with pytest.raises(C.NLTError):
        C.mul_forward(x, x) if False else C._ptr(x)


def test_bad_arguments_return_status_codes_without_a_gpu():
    """Null pointers / empty shapes are rejected by the argument checks before anything is launched, so these calls are
    safe on a machine with no GPU: every entry point answers NLT_ERR_BAD_ARG (-1) or NLT_ERR_UNSUPPORTED (-2)."""
    L = C.lib()
    n = None
    assert L.nlt_conv_forward(C.CONV_K2S1, C.ALGO_DIRECT, 0, n, 16, 16, n, 0, 0, 1, 4, 4, n, n, n, 16, n, 16, 1, 0.3, n, 0, 0, n) == -1
    assert L.nlt_stem_forward(n, n, n, n, n, n, 1, 1, 4, 4, 16, n, n, n, n, n, n, n) == -1
    assert L.nlt_head_forward(n, 4, 4, n, 32, 32, n, n, n, 1, 4, 4, n, n) == -1
    assert L.nlt_warp_forward(n, n, n, 1, 4, 4, 2, 2, n, n, n, n, n) == -1
    assert L.nlt_front_forward(n, n, n, n, n, 1, 1, 4, 4, n, 1, 0.3, n, n, n, n) == -1
    assert L.nlt_back_forward(n, n, n, 1, 2, 2, n, n, n, n, n, 0.3, n, n) == -1
    assert L.nlt_conv_tile_forward(C.CONV_K2S1, n, 16, 16, 1, 1, 4, 4, n, n, 32, 32, n, 32, n, 0, 1, 0.3, n) == -1
    assert L.nlt_conv_backward_weights_tiled(C.CONV_K2S1, n, 16, 16, n, 0, 0, 1, 4, 4, n, 16, 16, n, n, n, 0, n) == -1
    assert L.nlt_knn_indices(n, 0, n, 0, 1, n, n) == -1
    assert L.nlt_uv_index_map(n, n, 0, 1, 4, 4, 4, 0.0, n, n, n, n) == -1
    assert L.nlt_remap_bilinear_u8(n, 4, 4, 1, n, 0, 2, 4, 4, 1, n, n) == -1
    assert L.nlt_assemble_batch(n, n, n, n, n, n, 0, 0, 16, 0, n, n, n, n, n, n, n) == -1
    assert L.nlt_adam_amsgrad_step(n, n, n, n, n, 0, 1e-3, 0.9, 0.999, 1e-7, n) == -1
    assert L.nlt_chmix_bf16_forward(n, 0, 64, n, n, 64, 1, 0.3, n, n) == -1
    # sizes the kernels do not implement are "unsupported", not "bad"
    assert L.nlt_conv_tile_packed_floats(C.CONV_K2S1, 24, 32, 32) == -1 and L.nlt_chmix_bf16_packed_elems(48, 64) == -1
    assert L.nlt_wgrad_workspace_floats(C.CONV1X1, 5, 0, 1, 8, 8, 16) == -1


def test_binding_table_matches_the_header_prototypes_argument_for_argument():
    """Every prototype of include/nlt_hip.h against nlt_amd._capi.SIGNATURES: same number of parameters, pointers bound as
    c_void_p, float / double / long / int as themselves -- an ABI drift between the header and the ctypes table would
    otherwise only show up as garbage arguments on the GPU."""
    hdr = open(os.path.join(ROOT, 'include', 'nlt_hip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    protos = re.findall(r'\b(?:const\s+)?(int|long|const char\s*\*)\s+(nlt_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;', hdr, flags=re.S)
    assert len(protos) >= 60
    kinds = {ctypes.c_void_p: 'ptr', ctypes.c_int: 'int', ctypes.c_long: 'long', ctypes.c_float: 'float',
             ctypes.c_double: 'double', ctypes.c_char_p: 'ptr'}
    for ret, name, params in protos:
        res, args = C.SIGNATURES[name]
        params = params.strip()
        plist = [] if params in ('', 'void') else [p.strip() for p in params.split(',')]
        assert len(plist) == len(args), (name, len(plist), len(args))
        for p, a in zip(plist, args):
            want = 'ptr' if '*' in p else re.sub(r'\bconst\b', '', p).split()[0]
            want = {'nlt_map_dtype': 'int', 'unsigned': 'int'}.get(want, want)
            assert kinds[a] == want, (name, p, a)
        want_ret = 'ptr' if '*' in ret else ret
        assert kinds[res] == want_ret, (name, ret, res)


def test_native_tape_play_passes_arguments_like_a_direct_call():
    """nlt_tape_play (csrc/tape.hip) calls an entry point through a generic (32 integer-class, <= 4 float) signature.  Without a
    GPU the observable is the status code of the argument checks that run before anything is launched: for a 22-parameter entry
    (integer-class parameters well past the six register slots, two floats in between) every variation has to produce the status
    the direct ctypes call produces, and a failing call has to stop the replay at its index."""
    import ctypes
    L = C.lib()
    fn = L.nlt_conv_tile_backward_data
    P = 4096                                            # a non-null, 16-byte aligned fake pointer: checked, never dereferenced here

    def args(cpre=32, cout=32, tn=32, ldp=32, ldo=32, ldm=32, split_c=0, h=8):
        return (C.CONV_K2S1, P, ldp, cpre, 1, h, 8, P, cout, tn, P, ldo, P, ldm, 0.3, 0, split_c, None, None, 0.2, 0, None)
    cases = [dict(cpre=24), dict(tn=48), dict(ldp=16), dict(ldo=8), dict(ldm=4), dict(split_c=16), dict(cout=48), dict(h=0)]
    want = [fn(*args(**kw)) for kw in cases]
    assert set(want) <= {-1, -2} and -1 in want and -2 in want          # the checks were reached, both kinds
    calls = [C._describe(fn, args(**kw)) for kw in cases]
    assert all(c is not None and len(c[1]) == 20 and len(c[2]) == 2 for c in calls)
    for i, (c, w) in enumerate(zip(calls, want)):
        arr = (C._TapeCall * 1)()
        arr[0].fn, arr[0].n_float = c[0], len(c[2])
        for j, v in enumerate(c[1]):
            arr[0].iargs[j] = v
        for j, v in enumerate(c[2]):
            arr[0].fargs[j] = v
        failed = ctypes.c_int(-7)
        assert L.nlt_tape_play(arr, 1, ctypes.byref(failed)) == w and failed.value == 0, (i, cases[i])
    # a run stops at the first failing call
    ok = C._describe(L.nlt_stream_wait_event, (None, None))             # hipStreamWaitEvent(NULL, NULL) fails without a device too
    segs = C._compile([(fn, args(cpre=24))])
    assert segs[0][0] == 'native' and segs[0][2] == 1 and ok is not None
    # entries with double parameters stay Python calls
    assert C._describe(L.nlt_cosine_map, (None,) * 4 + (0.0, 0.0, 0.0, 0, None, None, None)) is None


def test_workspace_scope_and_thread_local_tape_bookkeeping():
    """Host-side state added for pipeline.RenderPipeline: the open launch tape belongs to the thread that opened it, the
    split-K scratch cache is keyed per plan (scope token) and forgotten with the plan."""
    import threading
    from nlt_amd import capi as C
    C.tape_begin()
    seen = []

    def other():
        seen.append(getattr(C._tls, 'tape', None))          # another thread: no tape open
        C.tape_begin(); C.tape_call(lambda: None); seen.append(len(C._tls.tape)); C.tape_abort()
    th = threading.Thread(target=other); th.start(); th.join()
    C.tape_call(lambda: None)
    t = C.tape_end('tag')
    assert seen == [None, 1] and len(t[0]) == 1 and C.tape_valid(t, 'tag') and not C.tape_valid(t, 'other')
    C.set_workspace_scope(1234)
    assert C._tls.scope == (1234 if C._WS_SCOPE else 0)
    C._splitk_ws[('dev', 1, 1234)] = object(); C._splitk_ws[('dev', 1, 99)] = object()
    C.drop_workspace_scope(1234)
    assert ('dev', 1, 1234) not in C._splitk_ws and ('dev', 1, 99) in C._splitk_ws
    del C._splitk_ws[('dev', 1, 99)]
    C.set_workspace_scope(0)
