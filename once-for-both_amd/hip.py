"""ctypes binding of libofb_hip.so (C ABI: include/ofb_hip.h).

PyTorch is used only as plumbing here: device buffers (`data_ptr()`), the current HIP stream
and the caching allocator.  All arithmetic happens inside the library's kernels.
"""
import ctypes as C
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('OFB_LIB_PATH') or os.path.join(_HERE, 'csrc', 'libofb_hip.so')      # OFB_LIB_PATH: lab A/B of two builds
_lib = None


class OfbError(RuntimeError):
    pass


ACT_NONE, ACT_GELU, ACT_DGELU, ACT_GELU_GRAD, ACT_MULAUX, ACT_GELU_GRAD_T, ACT_MULAUX_T = 0, 1, 2, 3, 4, 5, 6

# every symbol include/ofb_hip.h declares (tests/test_abi.py checks the .so exports exactly these)
SYMBOLS = [
    'ofb_gemm_h', 'ofb_gemm_h_workspace_bytes', 'ofb_gemm_h_colpart_rows', 'ofb_gemm_h_aux_t_floats', 'ofb_hformat_bytes', 'ofb_to_hformat', 'ofb_patchify_hformat', 'ofb_to_hformat_colsum', 'ofb_to_hformat_colsum_nb', 'ofb_to_hformat_multi', 'ofb_from_hformat', 'ofb_colsum_h', 'ofb_colsum_h_slabs', 'ofb_tune', 'ofb_gemm_h_rn_tiles',
    'ofb_splitk_reduce', 'ofb_prof_enable', 'ofb_prof_collect',
    'ofb_layernorm_fwd', 'ofb_layernorm_fwd_h', 'ofb_layernorm_bwd_blocks', 'ofb_layernorm_bwd', 'ofb_layernorm_bwd_h', 'ofb_layernorm_bwd_h_rn', 'ofb_colsum_slabs', 'ofb_colsum', 'ofb_colsum_multi',
    'ofb_scale_rows', 'ofb_gate_fold_bwd', 'ofb_amax', 'ofb_attention_fwd', 'ofb_attention_fwd_h', 'ofb_attention_bwd', 'ofb_attention_bwd_wgmax',
    'ofb_gates_fwd', 'ofb_gates_bwd', 'ofb_flops_loss',
    'ofb_embed_assemble_fwd', 'ofb_embed_assemble_chunks', 'ofb_embed_assemble_bwd', 'ofb_norm_targets', 'ofb_norm_targets_masked',
    'ofb_pmim_loss_fwd', 'ofb_pmim_loss_bwd', 'ofb_ls_cross_entropy', 'ofb_loss_mix', 'ofb_token_taps_fwd', 'ofb_token_taps_bwd', 'ofb_droppath_scales', 'ofb_scale_by_scalar', 'ofb_index_select', 'ofb_ema_update', 'ofb_adamw_step', 'ofb_adamw_step_dev', 'ofb_nonfinite_watch', 'ofb_multi_copy', 'ofb_upload', 'ofb_patch_mask', 'ofb_diag_mfma_peak', 'ofb_diag_cu_thief',
    'ofb_mixup_batch', 'ofb_mixup_targets', 'ofb_soft_cross_entropy', 'ofb_crop_resize_scratch_bytes', 'ofb_crop_resize_norm', 'ofb_random_erase',
    'ofb_randaug_layer', 'ofb_normalize_u8', 'ofb_jpeg_parse', 'ofb_jpeg_decode_coefficients', 'ofb_jpeg_plan_batch', 'ofb_jpeg_decode_batch', 'ofb_jpeg_decode_pixels',
]


def lib():
    """Load the HIP library or fail loudly (no CPU fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OfbError(f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                           '(hipcc --offload-arch=gfx950). once-for-both_amd has no CPU fallback.')
        _lib = C.CDLL(LIB_PATH)
        for s in SYMBOLS:
            getattr(_lib, s).restype = C.c_int
    return _lib


def check(rc, what):
    if rc != 0:
        raise OfbError(f'{what} failed with code {rc}' + (' (rejected arguments)' if rc < 0 else ' (hipError_t)'))


_OK_DTYPES = frozenset((torch.float32, torch.int64, torch.int32, torch.uint8, torch.bool))
_void_p = C.c_void_p


def ptr(t):
    # ~900 calls per search step: kept to two attribute reads and one ctypes object
    if t is None:
        return None
    if not t.is_cuda:
        raise OfbError('once-for-both_amd kernels need device tensors (no CPU fallback); got a CPU tensor')
    if t.dtype not in _OK_DTYPES:
        raise OfbError(f'unsupported dtype {t.dtype}')
    return _void_p(t.data_ptr())


def _dp(t):
    """ptr() for the per-call pointer writes of cached argument structs: the address as a plain int (ctypes converts it)"""
    if not t.is_cuda or t.dtype not in _OK_DTYPES:
        ptr(t)                                           # raises with the reason
    return t.data_ptr()


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_get_device = getattr(torch._C, '_cuda_getDevice', None)


def stream():
    """the current HIP stream of the current device as a void* (torch's raw-stream accessor: ~0.3 us instead of ~8 us for a
    torch.cuda.Stream object per launch)"""
    if _raw_stream is not None and _get_device is not None:
        return _void_p(_raw_stream(_get_device()))
    return _void_p(torch.cuda.current_stream().cuda_stream)


def _f32c(t, name):
    if t is not None and (t.dtype != torch.float32 or not t.is_contiguous()):
        raise OfbError(f'{name} must be a contiguous float32 tensor')
    return t


_gemm_ws = {}
_gemm_tl = threading.local()             # per thread: {call configuration: (argument struct, byref, partial rows, row-norm tiles, epoch, workspace bytes)}
_gemm_cfg_epoch = [0]                    # bumped by tune(): plans (workspace, partial counts) depend on the switches


def _workspace(device, nbytes):
    """one stream-K workspace per (device, stream): successive GEMMs on ONE stream may reuse it (stream order serialises the
    partial tiles), GEMMs on different streams (an EMA / eval forward on a side stream, user code) must not share it.
    A grown buffer replaces the old one; the caching allocator keeps the old block alive until the stream has passed it."""
    key = (device, _raw_stream(device.index if device.index is not None else torch.cuda.current_device())
           if _raw_stream is not None and isinstance(device, torch.device) else torch.cuda.current_stream(device).cuda_stream)
    ws = _gemm_ws.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        ws = torch.empty((nbytes + 3) // 4, device=device, dtype=torch.float32)
        _gemm_ws[key] = ws
    return ws


def gemm(A, B, C_out, M, N, K, lda, ldb, ldc, a_kc, b_kc, alpha=1.0, bias=None, colscale=None, rowscale=None, rs_div=1,
         resid=None, ldr=0, aux=None, ldaux=0, act=ACT_NONE, kscale=None, ks_div=1, a_colsum=None):
    """Convenience form for callers that hold plain f32 matrices (tests, scripts): one conversion pass per operand, then the
    plane GEMM.  a_kc / b_kc = 1: operand stored K-contiguous (A[m*lda+k], B[n*ldb+k]); 0: MN-contiguous.  The model path hands
    H-format tensors over directly (ops.py).  kscale (a_kc == 0): A's reduction rows scaled by kscale[k // ks_div]; a_colsum
    (a_kc == 0): column sums of the stored (scaled) A, i.e. the bias gradient beside a weight gradient."""
    if not a_kc and b_kc:
        raise OfbError('A^T * B^T is not a product of the path (ofb_gemm_h: OFB_ELIMIT)')
    if kscale is not None and a_kc:
        raise OfbError('kscale applies to an operand stored along the reduction (a_kc == 0)')
    Ap = to_hformat(A, M, K, lda) if a_kc else to_hformat(A, K, M, lda, rowscale=kscale, rs_div=ks_div)
    Bp = to_hformat(B, N, K, ldb) if b_kc else to_hformat(B, K, N, ldb)
    gemm_h(Ap, Bp, a_kc, b_kc, M, N, K, C_out=C_out, ldc=ldc, alpha=alpha, bias=bias, colscale=colscale, rowscale=rowscale,
           rs_div=rs_div, resid=resid, ldr=ldr, aux=aux, ldaux=ldaux, act=act)
    if a_colsum is not None:
        colsum(A, lda, K, M, a_colsum, rowscale=kscale, rs_div=ks_div)


# ---- H-format GEMM (csrc/gemm_h.hip): operands as two f16 planes of a power-of-two scaled copy + a device-side header ----------
class GemmHArgs(C.Structure):
    _fields_ = [
        ('A', C.c_void_p), ('B', C.c_void_p), ('a_kc', C.c_int32), ('b_kc', C.c_int32), ('a_ncb', C.c_int32), ('b_ncb', C.c_int32),
        ('M', C.c_int32), ('N', C.c_int32), ('K', C.c_int32),
        ('C', C.c_void_p), ('ldc', C.c_int32), ('Cp', C.c_void_p), ('c_ncb', C.c_int32),
        ('alpha', C.c_float), ('bias', C.c_void_p), ('colscale', C.c_void_p), ('rowscale', C.c_void_p), ('rs_div', C.c_int32),
        ('resid', C.c_void_p), ('ldr', C.c_int32), ('aux', C.c_void_p), ('ldaux', C.c_int32), ('act', C.c_int32),
        ('workspace', C.c_void_p), ('workspace_bytes', C.c_int64), ('colpart', C.c_void_p),
        ('aux_bound', C.c_float), ('out_bound', C.c_void_p), ('cbound_out', C.c_void_p),
        ('rn_gamma', C.c_void_p), ('rn_rowfac', C.c_void_p), ('rn_out', C.c_void_p),
    ]


_LN_ZERO_FILL = os.environ.get('OFB_LN_ZERO_FILL', '0') == '1'
_hbytes = {}                             # (R, C) -> ofb_hformat_bytes(R, C)


class HMat:
    """A matrix X[R][C] in H-format (csrc/hformat.h: a 256-byte device-side header {e, amax, row / column norm bounds}, then two
    f16 planes of X * 2^e in 4 x 16 granules).  `buf` is a uint8 device tensor of ofb_hformat_bytes(R, C) bytes."""
    __slots__ = ('buf', 'R', 'C', 'ncb')

    def __init__(self, R, C_, device, buf=None):
        self.R, self.C, self.ncb = int(R), int(C_), (int(C_) + 15) // 16
        if buf is None:
            n = _hbytes.get((R, C_))
            if n is None:                                    # (~150 buffers per step, a handful of shapes: the library is asked once per shape)
                f = lib().ofb_hformat_bytes
                f.restype = C.c_int64
                if len(_hbytes) > 4096:
                    _hbytes.clear()
                n = _hbytes[(R, C_)] = int(f(_i(R), _i(C_)))
            buf = torch.empty(n, device=device, dtype=torch.uint8)
        self.buf = buf

    @staticmethod
    def for_rows_written_by_kernel(R, C_, device):
        """planes that a producer kernel fills element by element: the padding rows / columns of the last granules, which a
        reduction along the rows / columns would read, are zeroed here"""
        pm = HMat(R, C_, device)
        if R % 16 or C_ % 16:
            pm.buf.zero_()
        return pm

    @staticmethod
    def for_rows_written_by_layernorm(R, C_, device):
        """planes the LayerNorm plane kernels fill: for an even width they write every column of the last 16-column granule themselves
        (zeros past the width: rowops.hip ln_p_store) and every 4-row group, so only the rows between R and the next multiple of 16 -
        which a reduction along the rows would read - need zeroing.  (The blanket zero fill of for_rows_written_by_kernel was 48 fills
        of 55 MB per finetune step at width 264: 0.47 ms.)"""
        pm = HMat(R, C_, device)
        if R % 16 or C_ % 2 or (_LN_ZERO_FILL and C_ % 16):            # (OFB_LN_ZERO_FILL=1: the blanket fill, for an A/B)
            pm.buf.zero_()
        return pm

    def to_f32(self):
        out = torch.empty(self.R, self.C, device=self.buf.device, dtype=torch.float32)
        check(lib().ofb_from_hformat(ptr(self.buf), _i(self.R), _i(self.C), ptr(out), _i(self.C), stream()), 'ofb_from_hformat')
        return out

    def header(self):
        """(e, amax, rn2sq, cn2sq) - a host copy for tests / diagnostics only (synchronises)"""
        h = self.buf[:16].cpu()
        f = h.view(torch.float32)
        return int(h.view(torch.int32)[0]), float(f[1]), float(f[2]), float(f[3])


TUNE_GEMM_MFMA, TUNE_GEMM_SCHED, TUNE_GEMM_TILE, TUNE_GEMM_T112, TUNE_GEMM_YIELD, TUNE_GEMM_DIRECT, TUNE_GEMM_CUS = 0, 1, 2, 3, 4, 5, 6


def tune(key, value):
    """run-time switch of a kernel variant (include/ofb_hip.h: OFB_TUNE_*): same results to rounding, for same-process A/B timing"""
    check(lib().ofb_tune(int(key), int(value)), 'ofb_tune')
    _gemm_cfg_epoch[0] += 1


def amax(x, out=None):
    """out[0] = max |x| on the device (no host sync): a bound for tensors whose producer left none"""
    if x.dtype != torch.float32 or not x.is_contiguous():
        raise OfbError('amax needs a contiguous float32 tensor')
    out = out if out is not None else torch.empty(1, device=x.device, dtype=torch.float32)
    check(lib().ofb_amax(ptr(x), C.c_int64(x.numel()), ptr(out), stream()), 'ofb_amax')
    return out


def to_hformat(x, R=None, Cc=None, ld=None, rowscale=None, rs_div=1, colsum_out=None, bound=None, into=None, colpart=False):
    """f32 [R][C] (row stride ld; default: the 2-D tensor's own shape / stride) -> HMat, optionally scaled per row by
    rowscale[r // rs_div].  colsum_out [C]: also receives the column sums of the (scaled) matrix from the same pass.
    bound: device scalar >= max |x * rowscale| (skips the statistics pass); into: existing planes (persistent weight planes)."""
    if R is None:
        R, Cc = x.shape
    if ld is None:
        ld = x.stride(0) if x.dim() == 2 else Cc
    if x.dtype != torch.float32:
        raise OfbError('to_hformat needs float32 input')
    pm = into if into is not None else HMat(R, Cc, x.device)
    if colsum_out is not None or colpart:
        slabs, ldp = int(lib().ofb_colsum_h_slabs(_i(R))), pm.ncb * 16
        part = torch.empty(slabs, ldp, device=x.device, dtype=torch.float32)
        if bound is not None and bound.numel() > 1:           # a vector of partial maxima (attention_bwd(..., wg_amax=...))
            check(lib().ofb_to_hformat_colsum_nb(ptr(x), _i(R), _i(Cc), _i(ld), ptr(pm.buf), ptr(rowscale), _i(rs_div), ptr(part), ptr(bound),
                                                 _i(bound.numel()), stream()), 'ofb_to_hformat_colsum_nb')
        else:
            check(lib().ofb_to_hformat_colsum(ptr(x), _i(R), _i(Cc), _i(ld), ptr(pm.buf), ptr(rowscale), _i(rs_div), ptr(part), ptr(bound),
                                              stream()), 'ofb_to_hformat_colsum')
        if colpart:                                            # (HMat, partial column sums [slabs][ldp], slabs): the consumer adds them up
            return pm, part, slabs
        colsum(part, ldp, slabs, Cc, colsum_out)
        return pm
    if bound is not None and bound.numel() > 1:
        # a vector of partial maxima without a consumer for the column sums (a qkv layer without bias): the entry point that reduces
        # the vector is the column-sum form; its partial sums land in a scratch buffer nobody reads
        slabs, ldp = int(lib().ofb_colsum_h_slabs(_i(R))), pm.ncb * 16
        part = torch.empty(slabs, ldp, device=x.device, dtype=torch.float32)
        check(lib().ofb_to_hformat_colsum_nb(ptr(x), _i(R), _i(Cc), _i(ld), ptr(pm.buf), ptr(rowscale), _i(rs_div), ptr(part), ptr(bound),
                                             _i(bound.numel()), stream()), 'ofb_to_hformat_colsum_nb')
        return pm
    check(lib().ofb_to_hformat(ptr(x), _i(R), _i(Cc), _i(ld), ptr(pm.buf), ptr(rowscale), _i(rs_div), ptr(bound), stream()), 'ofb_to_hformat')
    return pm


def patchify_hformat(imgs, patch):
    """[B][Cin][H][W] f32 images -> HMat of the patch matrix [B * L][Cin * patch^2] (conv-as-GEMM operand), no f32 copy in between"""
    if imgs.dtype != torch.float32 or not imgs.is_contiguous() or imgs.dim() != 4:
        raise OfbError('patchify_hformat needs a contiguous float32 [B][C][H][W] tensor')
    B, Cin, Hh, Ww = imgs.shape
    pm = HMat(B * (Hh // patch) * (Ww // patch), Cin * patch * patch, imgs.device)
    check(lib().ofb_patchify_hformat(ptr(imgs), _i(B), _i(Cin), _i(Hh), _i(Ww), _i(patch), ptr(pm.buf), stream()), 'ofb_patchify_hformat')
    return pm


def colsum_h(pm, out):
    """out[C] = column sums of an H-format matrix (two deterministic stages)."""
    slabs = int(lib().ofb_colsum_h_slabs(_i(pm.R)))
    ld = pm.ncb * 16
    part = torch.empty(slabs, ld, device=pm.buf.device, dtype=torch.float32)
    check(lib().ofb_colsum_h(ptr(pm.buf), _i(pm.R), _i(pm.C), ptr(part), stream()), 'ofb_colsum_h')
    colsum(part, ld, slabs, pm.C, out)


# ---- side stream for the weight-gradient products ------------------------------------------------------------------------
# dW = dY^T X is needed only by the optimizer, while dX = dY W sits on the critical path of backward.  The input-gradient launches
# leave CUs idle (394 tiles on 512 workgroup slots; HBM-bound LayerNorm / conversion kernels use no matrix pipe at all), so the
# weight-gradient launches (stream-K over all workgroups) go to a second HIP stream and fill those holes.  join_side() makes the
# current stream wait for them: at the end of backward (autograd callback), before the gate backward, before a gradient exchange.
SIDE_STREAM = os.environ.get('OFB_SIDE_STREAM', '1') != '0'
_side_streams, _side_keep, _side_dirty = {}, [], [False]


def ensure_side_stream(device, priority=None):
    """Creates the side stream of `device` (idempotent; `priority` given: replaces it, after joining the old one).
    HIP maps streams onto its hardware queues (GPU_MAX_HW_QUEUES, default 4) in creation order: with RCCL's own streams created first a
    normal-priority side stream lands on the main stream's hardware queue - the two then run one after the other and the
    weight-gradient overlap is lost (one rank, DeiT-S bs 128: 27.0 instead of 25.0 ms per step).  A HIGH-priority stream has a queue
    of its own whatever was created before it (25.4 ms with RCCL, 25.0 without), so that is the default for eager steps.  Inside a
    captured hipGraph the placement of parallel branches is a lottery (a high-priority branch, GPU_MAX_HW_QUEUES = 8 or merely other
    streams in the process make the replay of the bs-128 step 37 instead of 26 ms): engine.GraphedStep captures single-stream
    (scripts/lab/queue_matrix.sh has the table: queues x priority x eager / graph x plain / one-rank RCCL)."""
    device = torch.device(device)
    if device.index is None:
        device = torch.device('cuda', torch.cuda.current_device())
    side = _side_streams.get(device)
    if side is None or priority is not None:
        if side is not None:
            join_side()
            torch.cuda.current_stream(device).wait_stream(side)
        pr = int(os.environ.get('OFB_SIDE_PRIORITY', '-1')) if priority is None else int(priority)
        side = _side_streams[device] = torch.cuda.Stream(device=device, priority=pr)
    return side


class side_work:
    """with side_work(dev, keep=[tensors read inside]): launches go to the side stream, ordered after everything already queued on
    the current stream; `keep` tensors stay referenced until join_side() (the caching allocator must not recycle them earlier)."""

    def __init__(self, device, keep=()):
        self.device, self.keep = device, keep

    def __enter__(self):
        side = _side_streams.get(self.device)
        if side is None:
            side = ensure_side_stream(self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        _side_keep.extend(self.keep)
        _side_dirty[0] = True
        self.ctx = torch.cuda.stream(side)
        self.ctx.__enter__()
        return self

    def __exit__(self, *a):
        return self.ctx.__exit__(*a)


def join_side():
    """the current stream waits for all side-stream work issued so far"""
    if _side_dirty[0]:
        for dev, side in _side_streams.items():
            torch.cuda.current_stream(dev).wait_stream(side)
        _side_dirty[0] = False
        _side_keep.clear()


# H-format copies of the weights.  They live in a registry keyed by id(tensor) with a weak reference (never as attributes of the
# Parameter: torch pickles a Parameter's __dict__, so `torch.save(model)` - the reference's whole-object checkpoint format,
# search.py:671-740 - would write the planes into every checkpoint).  An entry dies with its tensor.  A copy is fresh while
#   * the weight epoch is unchanged: the optimizers / EMA / compress() / load_state_dict / the DP broadcast bump it after they
#     changed parameters through raw pointers or `.data`, and every model forward bumps it (begin_forward) - so a forward pass
#     NEVER trusts planes made before it started, whatever wrote the weights in between (`p.data.mul_()`, `p.data.copy_()`, the
#     reference's compress / resume idiom, moves neither `_version` nor any hook);
#   * the tensor's `_version`, data pointer and shape are unchanged (direct callers of the ops between two bumps).
# One copy feeds the forward, input-gradient and weight-gradient products.
_weight_epoch = 0


def bump_weight_epoch():
    global _weight_epoch
    _weight_epoch += 1
    _gw_jobs.clear()


_forward_hooks = []          # callables run at the start of every model forward (ops.begin_step: leftovers of a backward that raised)


def begin_forward():
    """called by the models at the start of every forward pass: all registered weights are re-converted (one multi-tensor launch
    pair, ~50 us for DeiT-S) by the first GEMM of the pass.  In a training step this coincides with the refresh the optimizer
    step asks for anyway."""
    bump_weight_epoch()
    for f in _forward_hooks:
        f()


class HformatJob(C.Structure):
    _fields_ = [('X', C.c_void_p), ('P', C.c_void_p), ('rowscale', C.c_void_p), ('R', C.c_int32), ('C', C.c_int32), ('ld', C.c_int32),
                ('pad_', C.c_int32)]


class _WEntry:
    """registry record of one weight: persistent plane buffers + what they were converted from"""
    __slots__ = ('ref', 'pm', 'epoch', 'version', 'ptr', 'shape', 'gw')

    def __init__(self, ref):
        self.ref, self.pm, self.epoch, self.version, self.ptr, self.shape, self.gw = ref, None, -1, -1, 0, None, None


# Every weight that has ever been asked for in P-format is registered.  The first request of a new epoch converts ALL registered
# weights that are stale in ONE multi-tensor launch (ofb_to_hformat_multi): ~70 launches per DeiT search step become one.  The
# job table is re-uploaded only when the set of (pointer, shape) entries changed.
_wp_reg = {}                 # id(W) -> _WEntry
_wp_table = [None, None, 0, 0]   # key tuple, device table (kept alive), max_R, max_C


def _drop_entry(key, r):
    e = _wp_reg.get(key)
    if e is not None and e.ref is r:
        del _wp_reg[key]


def _entry(W):
    """the registry record of tensor W (created on first use); None for objects that take no weak references"""
    import functools
    import weakref
    key = id(W)
    ent = _wp_reg.get(key)
    if ent is not None and ent.ref() is W:
        return ent
    try:
        ref = weakref.ref(W, functools.partial(_drop_entry, key))
    except TypeError:
        return None
    ent = _wp_reg[key] = _WEntry(ref)
    return ent


def _wp_fresh(W, ent):
    return (ent.pm is not None and ent.epoch == _weight_epoch and ent.version == W._version and ent.ptr == W.data_ptr()
            and ent.shape == tuple(W.shape))


_scratch = {}


def _multi_scratch(device, n_jobs):
    """two-stage maxima of a multi-tensor conversion: 64 floats per job"""
    device = torch.device(device)
    t = _scratch.get(device)
    if t is None or t.numel() < 64 * n_jobs:
        t = _scratch[device] = torch.empty(64 * max(n_jobs, 256), device=device, dtype=torch.float32)
    return t


def _wp_refresh_all(device):
    jobs, live = [], []
    for key, ent in list(_wp_reg.items()):
        W = ent.ref()
        if W is None:
            _wp_reg.pop(key, None)
            continue
        if ent.pm is None or ent.shape != tuple(W.shape):
            continue                                     # no planes yet / re-shaped in place: converted when it is asked for
        if W.device != device or _wp_fresh(W, ent) or not W.is_contiguous():
            continue
        live.append((W, ent))
        jobs.append((W.data_ptr(), ent.pm.buf.data_ptr(), ent.pm.R, ent.pm.C))
    if not jobs:
        return
    key = tuple(jobs)
    if _wp_table[0] != key:
        tab = (HformatJob * len(jobs))()
        for t, (x, pp, R, Cc) in zip(tab, jobs):
            t.X, t.P, t.rowscale, t.R, t.C, t.ld = x, pp, None, R, Cc, Cc
        dev_tab, host = upload_structs(tab, device)
        _wp_table[:] = [key, (dev_tab, host), max(j[2] for j in jobs), max(j[3] for j in jobs)]
    check(lib().ofb_to_hformat_multi(ptr(_wp_table[1][0]), _i(len(jobs)), _i(_wp_table[2]), _i(_wp_table[3]), ptr(_multi_scratch(device, len(jobs))),
                                     stream()), 'ofb_to_hformat_multi')
    for W, ent in live:
        ent.epoch, ent.version, ent.ptr = _weight_epoch, W._version, W.data_ptr()


_WP_MULTI = os.environ.get('OFB_WP_MULTI', '1') != '0'


def _param_of(W):
    """a reshaping view of a Parameter (decoder / patch-embed weights) is cached under the Parameter itself"""
    base = W._base
    if base is not None and base.is_contiguous() and base.numel() == W.numel() and base.data_ptr() == W.data_ptr():
        return base
    return W


def weight_h(W, shape2d=None):
    """H-format copy of a weight viewed as W[N][K] (shape2d: the 2-D view of a conv weight)."""
    if shape2d is None:
        shape2d = tuple(W.shape)
    W = _param_of(W)
    N, K = int(shape2d[0]), int(shape2d[1])
    ent = _entry(W)
    if ent is not None and _wp_fresh(W, ent) and (ent.pm.R, ent.pm.C) == (N, K):
        return ent.pm
    if not W.is_contiguous():
        raise OfbError('weight_h needs a contiguous weight')
    if ent is None:
        return to_hformat(W, N, K, K)                    # an object that takes no weak references: convert in place
    if not _WP_MULTI:
        ent.pm = to_hformat(W, N, K, K)
        ent.epoch, ent.version, ent.ptr, ent.shape = _weight_epoch, W._version, W.data_ptr(), tuple(W.shape)
        return ent.pm
    if ent.pm is None or ent.shape != tuple(W.shape) or (ent.pm.R, ent.pm.C) != (N, K):
        ent.pm, ent.shape, ent.epoch = HMat(N, K, W.device), tuple(W.shape), -1     # persistent planes; stale until converted
    _wp_refresh_all(W.device)
    if not _wp_fresh(W, ent):                            # e.g. another device's table was current: a single conversion
        to_hformat(W, N, K, K, into=ent.pm)
        ent.epoch, ent.version, ent.ptr = _weight_epoch, W._version, W.data_ptr()
    return ent.pm


def weight_registry_size():
    """number of live weights that hold H-format planes (tests)"""
    return sum(1 for e in _wp_reg.values() if e.ref() is not None and e.pm is not None)


# Gate-scaled weights g[n] * W[n][:] (the B operand of the gated layers' input-gradient products): the forward registers each
# (W, gate vector) pair, the first request in backward converts ALL pending pairs in one multi-tensor launch into per-weight
# persistent planes.  Entries live until the next bump of the weight epoch.
_gw_jobs = []                # (W, gvec, N, K)
_gw_table = [None, None, 0, 0]


def gated_register(W, gvec, N, K):
    if _WP_MULTI:
        _gw_jobs.append((_param_of(W), gvec, int(N), int(K)))


def _gw_fresh(W, gvec, ent):
    g = ent.gw if ent is not None else None
    return (g is not None and g[0] == _weight_epoch and g[1] == W._version and g[2] == gvec.data_ptr() and g[3] == gvec._version
            and g[5] == W.data_ptr())


def gated_weight_h(W, gvec, N, K):
    """H-format planes of gvec[n] * W[n][:] for W viewed as [N][K]"""
    W = _param_of(W)
    if not _WP_MULTI or not W.is_contiguous():
        return to_hformat(W, N, K, K, rowscale=gvec)
    ent = _entry(W)
    if ent is None:
        return to_hformat(W, N, K, K, rowscale=gvec)
    if _gw_fresh(W, gvec, ent) and (ent.gw[4].R, ent.gw[4].C) == (N, K):
        return ent.gw[4]
    todo, seen = [], set()
    for (W_, g_, N_, K_) in _gw_jobs + [(W, gvec, int(N), int(K))]:
        e_ = _entry(W_)
        if e_ is None or id(W_) in seen or W_.device != W.device or _gw_fresh(W_, g_, e_):
            continue
        seen.add(id(W_))
        pm = e_.gw[4] if e_.gw is not None and (e_.gw[4].R, e_.gw[4].C) == (N_, K_) else HMat(N_, K_, W_.device)
        todo.append((W_, g_, N_, K_, pm, e_))
    key = tuple((w_.data_ptr(), g_.data_ptr(), pm.buf.data_ptr(), n_, k_) for (w_, g_, n_, k_, pm, _) in todo)
    if _gw_table[0] != key:
        tab = (HformatJob * len(todo))()
        for t, (w_, g_, n_, k_, pm, _) in zip(tab, todo):
            t.X, t.P, t.rowscale, t.R, t.C, t.ld = w_.data_ptr(), pm.buf.data_ptr(), g_.data_ptr(), n_, k_, k_
        dev_tab, host = upload_structs(tab, W.device)
        _gw_table[:] = [key, (dev_tab, host), max(j[2] for j in todo), max(j[3] for j in todo)]
    check(lib().ofb_to_hformat_multi(ptr(_gw_table[1][0]), _i(len(todo)), _i(_gw_table[2]), _i(_gw_table[3]), ptr(_multi_scratch(W.device, len(todo))),
                                     stream()), 'ofb_to_hformat_multi')
    for (w_, g_, n_, k_, pm, e_) in todo:
        e_.gw = (_weight_epoch, w_._version, g_.data_ptr(), g_._version, pm, w_.data_ptr())
    return ent.gw[4]


def gemm_h(A, B, a_kc, b_kc, M, N, K, C_out=None, ldc=0, Cp=None, alpha=1.0, bias=None, colscale=None, rowscale=None, rs_div=1,
           resid=None, ldr=0, aux=None, ldaux=0, act=ACT_NONE, colsum_out=None, want_colpart=False, aux_bound=0.0, out_bound=None,
           cbound_out=None, rn=None):
    """C[M][N] (f32 and / or H-format) = A * B on H-format operands (HMat); a_kc / b_kc: reduction along the operand's columns.
    rn = (gamma [N], rowfac [M]): returns (per-tile row-norm maxima, sqrt(column tiles)) for ofb_layernorm_bwd_h_rn (include/ofb_hip.h).
    colsum_out [N]: also receives the column sums of the output (fused per-tile partial sums + one small reduction).
    want_colpart: return the per-tile partial column sums [rows][N] themselves (the consumer adds them up).
    aux_bound: bound of |aux| for the multiplying activations (default 1.13 = max gelu'); out_bound: device scalar that IS the
    bound of |output| for an H-format output (required with resid); cbound_out: device scalar that receives the bound of the
    f32 output (the exponent the attention kernels split it with)."""
    # ~150 calls per search step: everything that does not change from step to step - the argument struct with its scalars and
    # null pointers, the stream-K workspace size, the number of column-sum / row-norm partials - is kept per call configuration
    # (thread-local: the autograd engine runs backward on a thread of its own); a call then only writes the live pointers
    if Cp is not None and (Cp.R != M or Cp.C != N):
        raise OfbError('H-format output must be [M][N]')
    has_part = colsum_out is not None or want_colpart
    key = (a_kc, b_kc, M, N, K, A.ncb, B.ncb, ldc, -1 if Cp is None else Cp.ncb, alpha, rs_div, ldr, ldaux, act, aux_bound,
           C_out is None, bias is None, colscale is None, rowscale is None, resid is None, aux is None, out_bound is None,
           cbound_out is None, rn is None, has_part)
    cache = _gemm_tl.__dict__.get('c')
    if cache is None:
        cache = _gemm_tl.c = {}
    ent = cache.get(key)
    if ent is None or ent[4] != _gemm_cfg_epoch[0]:
        g = GemmHArgs()
        g.a_kc, g.b_kc, g.a_ncb, g.b_ncb = int(a_kc), int(b_kc), A.ncb, B.ncb
        g.M, g.N, g.K, g.ldc = M, N, K, ldc
        g.A, g.B, g.C = ptr(A.buf), ptr(B.buf), ptr(C_out)
        if Cp is not None:
            g.Cp, g.c_ncb = ptr(Cp.buf), Cp.ncb
        g.alpha, g.bias, g.colscale, g.rowscale, g.rs_div = alpha, ptr(bias), ptr(colscale), ptr(rowscale), rs_div
        g.resid, g.ldr, g.aux, g.ldaux, g.act = ptr(resid), ldr, ptr(aux), ldaux, act
        g.aux_bound, g.out_bound, g.cbound_out = aux_bound, ptr(out_bound), ptr(cbound_out)
        rows = int(lib().ofb_gemm_h_colpart_rows(C.byref(g))) if has_part else 0
        n_rn, nt = 0, C.c_int32(0)
        if rn is not None:
            g.rn_gamma, g.rn_rowfac = ptr(rn[0]), ptr(rn[1])
            n_rn = int(lib().ofb_gemm_h_rn_tiles(C.byref(g), C.byref(nt)))   # (0: a shape for the 96-column tile, which has no row-norm form)
            if n_rn <= 0:
                g.rn_gamma = g.rn_rowfac = None
        if has_part:
            g.colpart = 16                                   # (the plan looks at which outputs are asked for, not where they live)
        if n_rn > 0:
            g.rn_out = 16
        lib().ofb_gemm_h_workspace_bytes.restype = C.c_int64
        need = int(lib().ofb_gemm_h_workspace_bytes(C.byref(g)))
        if len(cache) > 4096:                                # (a long search re-shapes its products at every compress())
            cache.clear()
        ent = cache[key] = (g, C.byref(g), rows, (n_rn, float(nt.value) ** 0.5), _gemm_cfg_epoch[0], need)
    g, g_ref, rows, (n_rn, rn_fac), _, need = ent
    dev = A.buf.device
    g.A, g.B = _dp(A.buf), _dp(B.buf)
    if C_out is not None:
        g.C = _dp(C_out)
    if Cp is not None:
        g.Cp = _dp(Cp.buf)
    if bias is not None:
        g.bias = _dp(bias)
    if colscale is not None:
        g.colscale = _dp(colscale)
    if rowscale is not None:
        g.rowscale = _dp(rowscale)
    if resid is not None:
        g.resid = _dp(resid)
    if aux is not None:
        g.aux = _dp(aux)
    if out_bound is not None:
        g.out_bound = _dp(out_bound)
    if cbound_out is not None:
        g.cbound_out = _dp(cbound_out)
    part = None
    if has_part:
        part = torch.empty(rows, N, device=dev, dtype=torch.float32)
        g.colpart = part.data_ptr()
    rn_out = None
    if n_rn > 0:
        rn_out = (torch.empty(n_rn, device=dev, dtype=torch.float32), rn_fac)
        g.rn_gamma, g.rn_rowfac, g.rn_out = _dp(rn[0]), _dp(rn[1]), rn_out[0].data_ptr()
    if need > 0:
        ws = _workspace(dev, need)
        g.workspace, g.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    check(lib().ofb_gemm_h(g_ref, stream()), 'ofb_gemm_h')
    if want_colpart:
        return part
    if part is not None:
        colsum(part, N, part.shape[0], N, colsum_out)
    return rn_out


def aux_t(M, N, device):
    """buffer for the saved GELU derivative in T-layout (ACT_GELU_GRAD_T -> ACT_MULAUX_T; include/ofb_hip.h): private to the two
    GEMM launches that write and read it."""
    lib().ofb_gemm_h_aux_t_floats.restype = C.c_int64
    return torch.empty(int(lib().ofb_gemm_h_aux_t_floats(_i(M), _i(N))), device=device, dtype=torch.float32)


def aux_t_to_rows(aux, M, N):
    """T-layout -> [M][N] (tests / diagnostics only: the model path never looks inside)."""
    tm, tn = (M + 127) // 128, (N + 191) // 192
    a = aux.view(tm, tn, 2, 2, 4, 6, 4, 16, 4)            # tile m, tile n, wave row, wave col, row block, col block, row quad, col, row in quad
    a = a.permute(0, 2, 4, 6, 8, 1, 3, 5, 7).reshape(tm * 128, tn * 192)
    return a[:M, :N]


def splitk_reduce(ws, splits, count, out, accumulate=False):
    check(lib().ofb_splitk_reduce(ptr(ws), C.c_int32(splits), C.c_int64(count), ptr(out), C.c_int32(int(accumulate)),
                                  stream()), 'ofb_splitk_reduce')


def prof_enable(on):
    """on: False / 0 = off; True = every tag; an int = bit mask of the tags to bracket (bit 0: the GEMM)"""
    mask = 0x7fffffff if on is True else int(on)
    check(lib().ofb_prof_enable(C.c_int32(mask)), 'ofb_prof_enable')


def prof_collect(ntags=8):
    buf = (C.c_double * (ntags * 3))()
    check(lib().ofb_prof_collect(buf, C.c_int32(ntags)), 'ofb_prof_collect')
    return [(buf[3 * i], buf[3 * i + 1], buf[3 * i + 2]) for i in range(ntags)]


def _i(v):
    return C.c_int32(int(v))


def _f(v):
    return C.c_float(float(v))


def layernorm_fwd(x, gamma, beta, y, mean, rstd, rows, D, eps):
    check(lib().ofb_layernorm_fwd(ptr(x), ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(rstd), _i(rows), _i(D), _f(eps),
                                  stream()), 'ofb_layernorm_fwd')


def layernorm_fwd_h(x, gamma, beta, y, yP, mean, rstd, rows, D, eps):
    """LayerNorm rows as f32 `y` (may be None) and as H-format planes `yP` (HMat [rows][D]) from one pass."""
    check(lib().ofb_layernorm_fwd_h(ptr(x), ptr(gamma), ptr(beta), ptr(y), ptr(yP.buf), ptr(mean), ptr(rstd), _i(rows), _i(D),
                                    _f(eps), stream()), 'ofb_layernorm_fwd_h')


def layernorm_bwd_h(dy, x, gamma, mean, rstd, dres, dx, partials, dxP, rowscale, rs_div, rows, D):
    """LayerNorm backward that also writes dx * rowscale[row // rs_div] as planes `dxP`; partials: [blocks][3][D]."""
    check(lib().ofb_layernorm_bwd_h(ptr(dy), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(dres), ptr(dx), ptr(partials),
                                    ptr(dxP.buf), ptr(rowscale), _i(rs_div), _i(rows), _i(D), stream()), 'ofb_layernorm_bwd_h')


def layernorm_bwd_h_rn(dy, x, gamma, mean, rstd, dx, partials, dxP, rowscale, rs_div, rows, D, rn, rn_fac):
    """layernorm_bwd_h without its bound pass over dy: `rn` / `rn_fac` come from the GEMM that produced dy (gemm_h(rn=...))."""
    check(lib().ofb_layernorm_bwd_h_rn(ptr(dy), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(dx), ptr(partials), ptr(dxP.buf),
                                       ptr(rowscale), _i(rs_div), _i(rows), _i(D), ptr(rn), _i(rn.numel()), _f(rn_fac), stream()),
          'ofb_layernorm_bwd_h_rn')


def layernorm_bwd_blocks(rows):
    return int(lib().ofb_layernorm_bwd_blocks(_i(rows)))


def layernorm_bwd(dy, x, gamma, mean, rstd, dres, dx, partials, rows, D):
    check(lib().ofb_layernorm_bwd(ptr(dy), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(dres), ptr(dx), ptr(partials),
                                  _i(rows), _i(D), stream()), 'ofb_layernorm_bwd')


def colsum_slabs(M, N):
    return int(lib().ofb_colsum_slabs(_i(M), _i(N)))


def colsum(x, ld, M, N, out, rowscale=None, rs_div=1):
    slabs = colsum_slabs(M, N)
    scratch = torch.empty(slabs * N, device=x.device, dtype=torch.float32) if slabs > 1 else None
    check(lib().ofb_colsum(ptr(x), _i(ld), _i(M), _i(N), ptr(rowscale), _i(rs_div), ptr(out), ptr(scratch), stream()),
          'ofb_colsum')


# Column sums whose result only the optimizer (or the gradient exchange) reads - the LayerNorm weight / bias gradients and the
# output-bias gradients that ride on LayerNorm's backward partials - are not reduced one by one (two launches each) but queued and
# reduced together by ONE multi-job launch: at the end of backward (ops.py queues the callback), before a bucket of the
# data-parallel exchange leaves, before an optimizer step.
class ColsumJob(C.Structure):
    _fields_ = [('x', C.c_void_p), ('out', C.c_void_p), ('ld', C.c_int32), ('M', C.c_int32), ('N', C.c_int32), ('pad_', C.c_int32)]


_deferred = []               # (partials tensor, ld, M, N, out tensor)
_deferred_keep = [None]


def colsum_deferred(x, ld, M, N, out):
    _deferred.append((x, int(ld), int(M), int(N), out))


def flush_deferred():
    if not _deferred:
        return
    tab = (ColsumJob * len(_deferred))()
    for t, (x, ld, M, N, out) in zip(tab, _deferred):
        t.x, t.out, t.ld, t.M, t.N = x.data_ptr(), out.data_ptr(), ld, M, N
    dev = _deferred[0][0].device
    dev_tab, host = upload_structs(tab, dev)
    check(lib().ofb_colsum_multi(ptr(dev_tab), _i(len(_deferred)), _i(max(j[3] for j in _deferred)), stream()), 'ofb_colsum_multi')
    _deferred_keep[0] = (dev_tab, host, list(_deferred))       # inputs stay referenced until the next flush (stream order frees them)
    _deferred.clear()


def scale_rows(W, g, out, N, K):
    check(lib().ofb_scale_rows(ptr(W), ptr(g), ptr(out), _i(N), _i(K), stream()), 'ofb_scale_rows')


def gate_fold_bwd(dWraw, W, g, dbraw, b, dW, db, dg, N, K, dbraw_rows=1, fold=1):
    """dbraw: [dbraw_rows][N]; rows > 1: partial column sums (per image / per tile) that the kernel adds up itself.
    fold > 1: g is one gate of N / fold values tiled `fold` times (q | k | v); dg gets the N / fold sums over the groups"""
    if dbraw is not None and dbraw.numel() < dbraw_rows * N:
        raise OfbError('gate_fold_bwd: dbraw must hold dbraw_rows * N floats')
    if N % fold or dg.numel() < N // fold:
        raise OfbError('gate_fold_bwd: dg must hold N / fold floats')
    check(lib().ofb_gate_fold_bwd(ptr(dWraw), ptr(W), ptr(g), ptr(dbraw), _i(dbraw_rows), ptr(b), ptr(dW), ptr(db), ptr(dg), _i(N),
                                  _i(K), _i(fold), stream()), 'ofb_gate_fold_bwd')


def _check_lse(lse, B, N, H):
    if lse.numel() < 2 * B * H * N:
        raise OfbError('lse must hold 2 * B * H * N floats (log-sum-exp and its rounding residue)')


def attention_fwd(qkv, out, lse, B, N, H, dh, scale, qkv_bound=None):
    """qkv_bound: device scalar >= max |qkv| (the qkv GEMM's cbound_out); measured here when absent"""
    _check_lse(lse, B, N, H)
    qkv_bound = qkv_bound if qkv_bound is not None else amax(qkv)
    check(lib().ofb_attention_fwd(ptr(qkv), ptr(out), ptr(lse), _i(B), _i(N), _i(H), _i(dh), _f(scale), ptr(qkv_bound), stream()),
          'ofb_attention_fwd')
    return qkv_bound


def attention_fwd_h(qkv, out, outP, lse, B, N, H, dh, scale, qkv_bound=None):
    """forward that also writes the output as H-format planes (HMat [B*N][H*dh])"""
    if outP.R != B * N or outP.C != H * dh:
        raise OfbError('attention_fwd_h: output shapes')
    _check_lse(lse, B, N, H)
    qkv_bound = qkv_bound if qkv_bound is not None else amax(qkv)
    check(lib().ofb_attention_fwd_h(ptr(qkv), ptr(out), ptr(outP.buf), ptr(lse), _i(B), _i(N), _i(H), _i(dh), _f(scale), ptr(qkv_bound),
                                    stream()), 'ofb_attention_fwd_h')
    return qkv_bound


def attention_bwd(qkv, out, lse, dout, dqkv, B, N, H, dh, scale, qkv_bound=None, dout_bound=None, dqkv_amax=None, wg_amax=None):
    """qkv_bound: the bound the forward used; dout_bound: device scalar >= max |dout|; dqkv_amax: device scalar that receives
    max |dqkv| (the bound for its H-format copy); wg_amax [B * H] instead: one word per workgroup, written without atomics and without
    a memset node ahead of the launch - to_hformat(..., colsum_out=..., bound=wg_amax) reduces the vector itself"""
    _check_lse(lse, B, N, H)
    qkv_bound = qkv_bound if qkv_bound is not None else amax(qkv)
    dout_bound = dout_bound if dout_bound is not None else amax(dout)
    if wg_amax is not None:
        if dqkv_amax is not None or wg_amax.numel() != B * H or wg_amax.dtype != torch.float32:
            raise OfbError('attention_bwd: wg_amax must hold B * H floats (and excludes dqkv_amax)')
        check(lib().ofb_attention_bwd_wgmax(ptr(qkv), ptr(out), ptr(lse), ptr(dout), ptr(dqkv), _i(B), _i(N), _i(H), _i(dh), _f(scale),
                                            ptr(qkv_bound), ptr(dout_bound), ptr(wg_amax), stream()), 'ofb_attention_bwd_wgmax')
        return
    check(lib().ofb_attention_bwd(ptr(qkv), ptr(out), ptr(lse), ptr(dout), ptr(dqkv), _i(B), _i(N), _i(H), _i(dh), _f(scale),
                                  ptr(qkv_bound), ptr(dout_bound), ptr(dqkv_amax), stream()), 'ofb_attention_bwd')


# ---- gates / losses -------------------------------------------------------------------------------
class GateDesc(C.Structure):
    _fields_ = [
        ('alpha', C.c_void_p), ('score', C.c_void_p), ('g', C.c_void_p), ('wr', C.c_void_p), ('wm', C.c_void_p),
        ('prob', C.c_void_p), ('wsum', C.c_void_p), ('loss_alpha', C.c_void_p), ('dloss_dalpha', C.c_void_p),
        ('sig_partial', C.c_void_p), ('rank', C.c_void_p),
        ('H', C.c_int32), ('C', C.c_int32), ('A0', C.c_int32), ('A1', C.c_int32), ('kind', C.c_int32),
        ('w_p', C.c_float), ('norm_coef', C.c_float),
        ('head_thr', C.c_int32 * 8), ('chan_thr', C.c_int32 * 40), ('on', C.c_uint8 * 64),
    ]


class GateGrad(C.Structure):
    _fields_ = [('dg', C.c_void_p), ('dwr', C.c_void_p), ('dwm', C.c_void_p), ('dwsum', C.c_void_p), ('dspars', C.c_void_p),
                ('dalpha', C.c_void_p), ('dscore', C.c_void_p)]


class FlopsCfg(C.Structure):
    _fields_ = [('num_patches', C.c_int32), ('embed_dim', C.c_int32), ('num_heads', C.c_int32), ('head_dim', C.c_int32),
                ('hidden', C.c_int32), ('patch_area', C.c_int32), ('num_classes', C.c_int32), ('depth', C.c_int32),
                ('target', C.c_float), ('ln_dim', C.c_int32), ('active_heads', C.c_void_p), ('live_slot', C.c_void_p),
                ('wconst', C.c_void_p), ('n_live', C.c_int32), ('active_patches', C.c_void_p)]


class EmaTensor(C.Structure):
    _fields_ = [('ema', C.c_void_p), ('src', C.c_void_p), ('n', C.c_int64)]


class AdamwTensor(C.Structure):
    _fields_ = [('p', C.c_void_p), ('g', C.c_void_p), ('m', C.c_void_p), ('v', C.c_void_p), ('n', C.c_int64)]


# While a step is being captured into a hipGraph (engine.GraphedStep) the pointer tables the step uploads (gate descriptors, AdamW
# tensor lists) are staged in ONE pinned arena that the capture owns: each upload becomes a memcpy node that every replay re-reads
# from the same host address.  (torch's pinned allocator and its non_blocking copies record / query events, which a capture does not
# allow - least of all from the autograd thread.)
_arena = None


def begin_capture_arena(nbytes=1 << 20):
    global _arena
    _arena = dict(host=torch.empty(nbytes, dtype=torch.uint8).pin_memory(), off=0)
    return _arena


def end_capture_arena():
    global _arena
    _arena = None


def upload_structs(array, device):
    """ctypes struct array -> device byte tensor (pinned staging, async on the current stream)."""
    if torch.device(device).type != 'cuda':
        raise OfbError('once-for-both_amd kernels need device tensors (no CPU fallback); got a CPU tensor')
    raw = bytes(array)
    if _arena is not None:
        n, off = len(raw), (_arena['off'] + 63) & ~63
        if off + n > _arena['host'].numel():
            raise OfbError('capture arena exhausted')
        host = _arena['host'][off:off + n]
        C.memmove(host.data_ptr(), raw, n)
        _arena['off'] = off + n
        dev = torch.empty(n, dtype=torch.uint8, device=device)
        check(lib().ofb_upload(ptr(dev), C.c_void_p(host.data_ptr()), C.c_int64(n), stream()), 'ofb_upload')
        return dev, host
    # pinned block from torch's caching host allocator, filled by memmove: `frombuffer(...).pin_memory()` copies with ATen's parallel
    # copy, which above 32 K elements wakes every OpenMP thread of the process (128 spinning threads on a 16-CPU cgroup share got
    # the whole process throttled for ~90 ms at a time: JPEG job tables, scripts/lab/time_jpeg_stages.py)
    host = torch.empty(len(raw), dtype=torch.uint8, pin_memory=True)
    C.memmove(host.data_ptr(), raw, len(raw))
    return host.to(device, non_blocking=True), host


def gates_fwd(descs_dev, n, max_elems, entropy, var, norm, spars_out, spars_per_module):
    check(lib().ofb_gates_fwd(ptr(descs_dev), _i(n), _i(max_elems), _i(entropy), _i(var), _i(norm), ptr(spars_out),
                              ptr(spars_per_module), stream()), 'ofb_gates_fwd')


def gates_bwd(descs_dev, grads_dev, n):
    check(lib().ofb_gates_bwd(ptr(descs_dev), ptr(grads_dev), _i(n), stream()), 'ofb_gates_bwd')


def flops_loss(wsum, cfg, out3, dwsum):
    check(lib().ofb_flops_loss(ptr(wsum), C.byref(cfg), ptr(out3), ptr(dwsum), stream()), 'ofb_flops_loss')


# ---- embed / PMIM / CE / AdamW --------------------------------------------------------------------
def embed_assemble_fwd(conv, g, pos, cls, mask_token, mask, tokens, B, L, D):
    check(lib().ofb_embed_assemble_fwd(ptr(conv), ptr(g), ptr(pos), ptr(cls), ptr(mask_token), ptr(mask), ptr(tokens), _i(B),
                                       _i(L), _i(D), stream()), 'ofb_embed_assemble_fwd')


def embed_assemble_chunks(B):
    return int(lib().ofb_embed_assemble_chunks(_i(B)))


def embed_assemble_bwd(dtokens, conv, g, pos, cls, mask_token, mask, dconv, ppos, pg, pmt, B, L, D):
    check(lib().ofb_embed_assemble_bwd(ptr(dtokens), ptr(conv), ptr(g), ptr(pos), ptr(cls), ptr(mask_token), ptr(mask),
                                       ptr(dconv), ptr(ppos), ptr(pg), ptr(pmt), _i(B), _i(L), _i(D), stream()),
          'ofb_embed_assemble_bwd')


def norm_targets(imgs, out, s1, s2, planes, H, W, k=47):
    check(lib().ofb_norm_targets(ptr(imgs), ptr(out), ptr(s1), ptr(s2), _i(planes), _i(H), _i(W), _i(k), stream()),
          'ofb_norm_targets')


def norm_targets_masked(imgs, ids, out, B, Cc, L, P, H, W, k=47):
    check(lib().ofb_norm_targets_masked(ptr(imgs), ptr(ids), _i(ids.numel()), ptr(out), _i(B), _i(Cc), _i(L), _i(P), _i(H), _i(W), _i(k),
                                        stream()), 'ofb_norm_targets_masked')


def pmim_loss_fwd(rec, targets, mask, ids, n_rows, partial, out2, B, L, P, Cc):
    check(lib().ofb_pmim_loss_fwd(ptr(rec), ptr(targets), ptr(mask), ptr(ids), _i(n_rows), ptr(partial), ptr(out2), _i(B), _i(L),
                                  _i(P), _i(Cc), stream()), 'ofb_pmim_loss_fwd')


def pmim_loss_bwd(rec, targets, mask, ids, n_rows, out2, upstream, drec, B, L, P, Cc):
    check(lib().ofb_pmim_loss_bwd(ptr(rec), ptr(targets), ptr(mask), ptr(ids), _i(n_rows), ptr(out2), ptr(upstream), ptr(drec),
                                  _i(B), _i(L), _i(P), _i(Cc), stream()), 'ofb_pmim_loss_bwd')


def ls_cross_entropy(logits, labels, row_loss, loss, grad, B, Cn, smoothing):
    check(lib().ofb_ls_cross_entropy(ptr(logits), ptr(labels), ptr(row_loss), ptr(loss), ptr(grad), _i(B), _i(Cn),
                                     _f(smoothing), stream()), 'ofb_ls_cross_entropy')


def token_taps_fwd(stream_rows, patch_ids, n_ids, B, T, D, cls_out, z_out):
    check(lib().ofb_token_taps_fwd(ptr(stream_rows), ptr(patch_ids), _i(n_ids), _i(B), _i(T), _i(D), ptr(cls_out), ptr(z_out), stream()),
          'ofb_token_taps_fwd')


def token_taps_bwd(dcls, dz, patch_ids, n_ids, B, T, D, dstream):
    check(lib().ofb_token_taps_bwd(ptr(dcls), ptr(dz), ptr(patch_ids), _i(n_ids), _i(B), _i(T), _i(D), ptr(dstream), stream()),
          'ofb_token_taps_bwd')


def droppath_scales(u, keep, out, R, B):
    check(lib().ofb_droppath_scales(ptr(u), ptr(keep), ptr(out), _i(R), _i(B), stream()), 'ofb_droppath_scales')


def loss_mix(base, spars3, flops, dec, w, out3):
    """out3 = (arch, base / dec, base + arch + (base / dec) * dec) from device scalars (any of the inputs None): include/ofb_hip.h."""
    check(lib().ofb_loss_mix(ptr(base), ptr(spars3), ptr(flops), ptr(dec), _f(w[0]), _f(w[1]), _f(w[2]), _f(w[3]), ptr(out3), stream()),
          'ofb_loss_mix')


def scale_by_scalar(x, scalar_dev, out, n):
    check(lib().ofb_scale_by_scalar(ptr(x), ptr(scalar_dev), ptr(out), C.c_int64(n), stream()), 'ofb_scale_by_scalar')


def ema_update(table_dev, n_tensors, max_numel, decay):
    check(lib().ofb_ema_update(ptr(table_dev), _i(n_tensors), C.c_int64(max_numel), C.c_float(decay), C.c_float(1.0 - decay),
                               _skip_ptr(table_dev.device), stream()), 'ofb_ema_update')


# ---- non-finite loss watch (reference engine.py:146-150 stops before backward when the loss is not finite) -----------------------
# One int32 counter per device: watch_nonfinite(loss) bumps it on the device when the loss is NaN / inf; from then on the AdamW and
# EMA kernels (which receive the counter as `skip`) change nothing, so no update is ever applied after a non-finite loss although
# the host looks at the counter only at its print points.
_nonfinite = {}


def nonfinite_flag(device, create=True):
    device = torch.device(device)
    if device.type == 'cuda' and device.index is None:
        device = torch.device('cuda', torch.cuda.current_device())
    t = _nonfinite.get(device)
    if t is None and create:
        t = _nonfinite[device] = torch.zeros(1, device=device, dtype=torch.int32)
    return t


def _skip_ptr(device):
    t = nonfinite_flag(device, create=False)
    return None if t is None else _void_p(t.data_ptr())


def watch_nonfinite(values):
    """values: f32 device tensor (the step's total loss): counts a non-finite entry on the device, no host sync"""
    v = values.detach().reshape(-1)
    if v.dtype != torch.float32 or not v.is_contiguous():
        v = v.float().contiguous()
    check(lib().ofb_nonfinite_watch(ptr(v), _i(v.numel()), ptr(nonfinite_flag(v.device)), stream()), 'ofb_nonfinite_watch')


def reset_nonfinite(device=None):
    if device is not None:
        device = torch.device(device)
        if device.type == 'cuda' and device.index is None:              # the registry's keys carry an index (nonfinite_flag)
            device = torch.device('cuda', torch.cuda.current_device())
    for d, t in _nonfinite.items():
        if device is None or d == device:
            t.zero_()


class CopyJob(C.Structure):
    _fields_ = [('src', C.c_void_p), ('dst', C.c_void_p), ('n', C.c_int64)]


def multi_copy(pairs, device):
    """pairs: [(src tensor or None, dst tensor)] contiguous f32, same numel: dst <- src (or zeros) in ONE launch"""
    tab = (CopyJob * len(pairs))()
    maxn = 0
    for t, (src, dst) in zip(tab, pairs):
        if src is not None and (src.dtype != torch.float32 or not src.is_contiguous() or src.numel() != dst.numel()):
            raise OfbError('multi_copy: sources must be contiguous float32 tensors of the destination size')
        t.src, t.dst, t.n = (None if src is None else src.data_ptr()), dst.data_ptr(), dst.numel()
        maxn = max(maxn, dst.numel())
    dev_tab, host = upload_structs(tab, device)
    check(lib().ofb_multi_copy(ptr(dev_tab), _i(len(pairs)), C.c_int64(maxn), stream()), 'ofb_multi_copy')
    return dev_tab, host


def index_select(t, index, dim):
    """t.index_select(dim, index) on the device through ofb_index_select (compress() surgery).  `index`: integer tensor or
    list; returns a new contiguous tensor.  Out-of-range indices raise OfbError."""
    if not t.is_cuda:
        raise OfbError('ofb_index_select needs a device tensor (no CPU fallback)')
    t = t if t.is_contiguous() else t.contiguous()
    dim = dim % t.dim()
    idx = torch.as_tensor(index).reshape(-1)
    on_host = not idx.is_cuda
    if on_host and idx.numel() and (int(idx.min()) < 0 or int(idx.max()) >= t.shape[dim]):
        raise OfbError('ofb_index_select: index out of range')
    idx = idx.to(device=t.device, dtype=torch.int32)
    shape = list(t.shape)
    outer = 1
    for s in shape[:dim]:
        outer *= s
    inner = 1
    for s in shape[dim + 1:]:
        inner *= s
    n_src, n_idx = shape[dim], idx.numel()
    out = torch.empty(shape[:dim] + [n_idx] + shape[dim + 1:], device=t.device, dtype=torch.float32)
    bad = None if on_host else torch.zeros(1, device=t.device, dtype=torch.int32)     # host-built indices were checked above
    check(lib().ofb_index_select(ptr(t), ptr(idx), ptr(out), C.c_int64(outer), C.c_int64(n_src), C.c_int64(n_idx), C.c_int64(inner),
                                 ptr(bad), stream()), 'ofb_index_select')
    if bad is not None and int(bad.item()):
        raise OfbError('ofb_index_select: index out of range')
    return out


def adamw_step(table_dev, n_tensors, max_numel, lr, beta1, beta2, eps, wd, step):
    check(lib().ofb_adamw_step(ptr(table_dev), _i(n_tensors), C.c_int64(max_numel), _f(lr), _f(beta1), _f(beta2), _f(eps),
                               _f(wd), _i(step), _skip_ptr(table_dev.device), stream()), 'ofb_adamw_step')


def adamw_step_dev(table_dev, n_tensors, max_numel, hyper_dev, beta1, beta2, eps, wd):
    check(lib().ofb_adamw_step_dev(ptr(table_dev), _i(n_tensors), C.c_int64(max_numel), ptr(hyper_dev), _f(beta1), _f(beta2), _f(eps),
                                   _f(wd), _skip_ptr(table_dev.device), stream()), 'ofb_adamw_step_dev')


def patch_mask(noise, mask, B, L, len_keep, masked_ids=None):
    check(lib().ofb_patch_mask(ptr(noise), ptr(mask), ptr(masked_ids), _i(B), _i(L), _i(len_keep), stream()), 'ofb_patch_mask')


def diag_mfma_peak(out, blocks, iters):
    check(lib().ofb_diag_mfma_peak(ptr(out), _i(blocks), _i(iters), stream()), 'ofb_diag_mfma_peak')


class MixParam(C.Structure):
    _fields_ = [('lam', C.c_float), ('one_minus_lam', C.c_float), ('use_cutmix', C.c_int32),
                ('yl', C.c_int32), ('yh', C.c_int32), ('xl', C.c_int32), ('xh', C.c_int32)]


class CropParam(C.Structure):
    _fields_ = [('offset', C.c_int64), ('src_h', C.c_int32), ('src_w', C.c_int32), ('top', C.c_int32), ('left', C.c_int32),
                ('height', C.c_int32), ('width', C.c_int32), ('flip', C.c_int32), ('cubic', C.c_int32)]


def mixup_batch(x, params_dev, B, Cc, H, W):
    check(lib().ofb_mixup_batch(ptr(_f32c(x, 'x')), ptr(params_dev), _i(B), _i(Cc), _i(H), _i(W), stream()), 'ofb_mixup_batch')


def mixup_targets(labels, params_dev, out, B, ncls, on_value, off_value):
    if labels.dtype != torch.int64 or not labels.is_contiguous():
        raise OfbError('labels must be a contiguous int64 tensor')
    check(lib().ofb_mixup_targets(ptr(labels), ptr(params_dev), ptr(out), _i(B), _i(ncls), _f(on_value), _f(off_value), stream()),
          'ofb_mixup_targets')


def soft_cross_entropy(logits, target, row_loss, loss, grad, B, Cn):
    check(lib().ofb_soft_cross_entropy(ptr(_f32c(logits, 'logits')), ptr(_f32c(target, 'target')), ptr(row_loss), ptr(loss), ptr(grad),
                                       _i(B), _i(Cn), stream()), 'ofb_soft_cross_entropy')


def crop_resize_scratch_bytes(B, S, max_src_h):
    f = lib().ofb_crop_resize_scratch_bytes
    f.restype = C.c_int64
    return int(f(_i(B), _i(S), _i(max_src_h)))


def crop_resize_norm(src_u8, params_dev, B, S, max_src_h, mean, std, out, out_u8, scratch):
    if src_u8.dtype != torch.uint8 or scratch.dtype != torch.uint8:
        raise OfbError('source / scratch must be uint8 device tensors')
    m3, s3 = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
    check(lib().ofb_crop_resize_norm(ptr(src_u8), ptr(params_dev), _i(B), _i(S), _i(max_src_h), m3, s3, ptr(out), ptr(out_u8), ptr(scratch),
                                     stream()), 'ofb_crop_resize_norm')


class EraseParam(C.Structure):
    _fields_ = [('top', C.c_int32), ('left', C.c_int32), ('h', C.c_int32), ('w', C.c_int32)]


def random_erase(x, params_dev, B, Cc, H, W, seed):
    check(lib().ofb_random_erase(ptr(_f32c(x, 'x')), ptr(params_dev), _i(B), _i(Cc), _i(H), _i(W), C.c_uint64(int(seed) & (2 ** 64 - 1)),
                                 stream()), 'ofb_random_erase')


class AugOp(C.Structure):
    _fields_ = [('op', C.c_int32), ('iarg', C.c_int32), ('farg', C.c_float), ('fill', C.c_int32 * 3), ('m', C.c_double * 6)]


def randaug_layer(src_u8, dst_u8, ops_dev, B, H, W, hist, lsum):
    if src_u8.dtype != torch.uint8 or dst_u8.dtype != torch.uint8 or hist.dtype != torch.int32 or lsum.dtype != torch.int64:
        raise OfbError('randaug_layer: uint8 images, int32 histogram scratch, int64 luma scratch')
    check(lib().ofb_randaug_layer(ptr(src_u8), ptr(dst_u8), ptr(ops_dev), _i(B), _i(H), _i(W), ptr(hist), ptr(lsum), stream()), 'ofb_randaug_layer')


def normalize_u8(src_u8, out, B, H, W, mean, std):
    m3, s3 = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
    check(lib().ofb_normalize_u8(ptr(src_u8), ptr(_f32c(out, 'out')), _i(B), _i(H), _i(W), m3, s3, stream()), 'ofb_normalize_u8')


# ---- JPEG decode (csrc/jpeg.hip): host entropy stage + device IDCT / upsampling / colour -------------------------------------
class JpegInfo(C.Structure):
    _fields_ = [('width', C.c_int32), ('height', C.c_int32), ('ncomp', C.c_int32), ('hs', C.c_int32 * 3), ('vs', C.c_int32 * 3),
                ('hmax', C.c_int32), ('vmax', C.c_int32), ('mcu_x', C.c_int32), ('mcu_y', C.c_int32), ('blocks_w', C.c_int32 * 3),
                ('blocks_h', C.c_int32 * 3), ('pad_', C.c_int32), ('coef_off', C.c_int64 * 3), ('coef_count', C.c_int64),
                ('quant', (C.c_uint16 * 64) * 3)]


class JpegJob(C.Structure):
    _fields_ = [('width', C.c_int32), ('height', C.c_int32), ('ncomp', C.c_int32), ('hs', C.c_int32 * 3), ('vs', C.c_int32 * 3),
                ('hmax', C.c_int32), ('vmax', C.c_int32), ('blocks_w', C.c_int32 * 3), ('blocks_h', C.c_int32 * 3), ('pad_', C.c_int32),
                ('coef_off', C.c_int64 * 3), ('plane_off', C.c_int64 * 3), ('out_off', C.c_int64), ('quant', (C.c_uint16 * 64) * 3)]


def jpeg_parse(data):
    """frame header of a JPEG file (bytes) -> JpegInfo; OfbError for files outside the decoder's scope (progressive, CMYK, ...)"""
    info = JpegInfo()
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    check(lib().ofb_jpeg_parse(buf, C.c_int64(len(data)), C.byref(info)), 'ofb_jpeg_parse')
    return info, buf


def jpeg_decode_coefficients(buf, nbytes, info, out_ptr):
    """host Huffman stage: writes info.coef_count int16 at out_ptr (the ctypes call releases the GIL: run it on worker threads)"""
    check(lib().ofb_jpeg_decode_coefficients(buf, C.c_int64(nbytes), C.byref(info), C.c_void_p(out_ptr)), 'ofb_jpeg_decode_coefficients')


class JpegBatch:
    """the host-side plan of a batch of JPEG files (ofb_jpeg_plan_batch): per-file info and device job records, the buffer sizes"""
    __slots__ = ('n', 'files', 'nbytes', 'infos', 'jobs', 'coef_total', 'plane_total', 'out_total', 'max_blocks', 'max_w', 'max_h', '_blobs')


def jpeg_plan_batch(blobs):
    n = len(blobs)
    if n == 0:
        raise OfbError('jpeg_plan_batch: empty batch')
    blobs = [b if isinstance(b, bytes) else bytes(b) for b in blobs]
    pb = JpegBatch()
    pb.n, pb._blobs = n, blobs
    pb.files = (C.c_char_p * n)(*blobs)                       # pointers into the bytes objects (kept alive by _blobs)
    pb.nbytes = (C.c_int64 * n)(*[len(b) for b in blobs])
    pb.infos, pb.jobs = (JpegInfo * n)(), (JpegJob * n)()
    totals = (C.c_int64 * 6)()
    check(lib().ofb_jpeg_plan_batch(pb.files, pb.nbytes, _i(n), pb.infos, pb.jobs, totals), 'ofb_jpeg_plan_batch')
    pb.coef_total, pb.plane_total, pb.out_total, pb.max_blocks, pb.max_w, pb.max_h = [int(v) for v in totals]
    return pb


def jpeg_decode_batch(pb, coef_ptr, threads):
    """host Huffman stage of the whole batch on `threads` native threads (the ctypes call releases the GIL); coef_ptr: pb.coef_total
    int16 of (pinned) host memory"""
    check(lib().ofb_jpeg_decode_batch(pb.files, pb.nbytes, _i(pb.n), pb.infos, pb.jobs, C.c_void_p(coef_ptr), _i(max(1, int(threads)))),
          'ofb_jpeg_decode_batch')


def jpeg_decode_pixels(jobs_dev, n, max_blocks, max_w, max_h, coef_dev, planes_dev, out_dev):
    if coef_dev.dtype != torch.int16 or planes_dev.dtype != torch.uint8 or out_dev.dtype != torch.uint8:
        raise OfbError('jpeg_decode_pixels: int16 coefficients, uint8 planes / pixels')
    check(lib().ofb_jpeg_decode_pixels(ptr(jobs_dev), _i(n), _i(max_blocks), _i(max_w), _i(max_h), _void_p(coef_dev.data_ptr()),
                                       ptr(planes_dev), ptr(out_dev), stream()), 'ofb_jpeg_decode_pixels')
