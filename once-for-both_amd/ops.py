"""torch.autograd.Function wrappers around the C-ABI kernels (hip.py).

Each Function is one fused op group of SURVEY.md 2.3 with a hand-written backward; torch only
allocates buffers and orders the graph.  All tensors are contiguous fp32 device tensors.
"""
import ctypes as C
import os

import torch

from . import hip
from .dp import grad_slot


def _new(like, *shape):
    return torch.empty(shape, device=like.device, dtype=torch.float32)


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


# ---------------------------------------------------------------------------------------------------
# H-format path (csrc/gemm_h.hip, csrc/hformat.h): every GEMM operand is handed over as two pre-split f16 planes of a power-of-two
# scaled copy, its exponent in a device-side header.  Producers on the path (LayerNorm, the GELU epilogue of fc1, ...) attach the
# H-format copy of their output to the tensor object (`_ofb_p`); anything that arrives without one is converted by ofb_to_hformat
# (a statistics pass + a conversion pass).  Weights are converted once per optimizer step (hip.weight_h).
# ---------------------------------------------------------------------------------------------------
def _P(t, M, K):
    """H-format [M][K] copy of an activation (cached on the tensor object by its producer, else converted now)"""
    pm = getattr(t, '_ofb_p', None)
    if pm is not None and pm.R == M and pm.C == K:
        return pm
    return hip.to_hformat(t, M, K, K)


def _pm(buf, R, C):
    return hip.HMat(R, C, buf.device, buf=buf)


def p_linear_fwd(xP, M, K, WP, b, colscale=None, act=hip.ACT_NONE, aux=None, rowscale=None, rs_div=1, resid=None, want_f32=True,
                 want_p=False, cbound_out=None):
    """y[M,N] = x[M,K] @ W[N,K]^T (+ fused epilogue), WP = hip.weight_h(W); returns (f32 y or None, H-format y or None).
    cbound_out: device scalar that receives an upper bound of |y| (for a consumer that splits y itself: attention)"""
    N, dev = WP.R, WP.buf.device
    y = torch.empty(M, N, device=dev, dtype=torch.float32) if want_f32 else None
    yP = hip.HMat(M, N, dev) if want_p else None
    hip.gemm_h(xP, WP, 1, 1, M, N, K, C_out=y, ldc=N, Cp=yP, bias=b, colscale=colscale, act=act, aux=aux, ldaux=N,
               rowscale=rowscale, rs_div=rs_div, resid=resid, ldr=N, cbound_out=cbound_out)
    return y, yP


def p_linear_bwd_input(dyP, M, N, WP, K, resid=None, act=hip.ACT_NONE, aux=None, want_f32=True, want_p=False, colsum_out=None,
                       want_colpart=False, cbound_out=None, ln=None):
    """dX[M,K] = dY[M,N] @ W[N,K]: dY reduced along its columns (KC), W along its rows (KR): the SAME H-format copy of W as forward.
    want_colpart: third return value = the per-tile partial column sums of dX [rows][K]
    ln = (gamma, rstd) of the LayerNorm whose output x was: the epilogue also leaves what that LayerNorm's backward needs to bound its
    result (parked under dX's identity for it: _put_rn)"""
    dx = torch.empty(M, K, device=dyP.buf.device, dtype=torch.float32) if want_f32 else None
    dxP = hip.HMat(M, K, dyP.buf.device) if want_p else None
    use_rn = ln is not None and want_f32 and not want_colpart and colsum_out is None and _RN_HANDOVER
    part = hip.gemm_h(dyP, WP, 1, 0, M, K, N, C_out=dx, ldc=K, Cp=dxP, resid=resid, ldr=K, act=act, aux=aux, ldaux=K, colsum_out=colsum_out,
                      want_colpart=want_colpart, cbound_out=cbound_out, rn=ln if use_rn else None)
    if use_rn and part is not None:
        _put_rn(dx, part)
    return (dx, dxP, part) if want_colpart else (dx, dxP)


def p_linear_bwd_weight(dyP, xP, M, N, K, out=None):
    """dW[N,K] = dY[M,N]^T @ X[M,K]: both reduced along their rows (the tokens)"""
    dW = out if out is not None else torch.empty(N, K, device=dyP.buf.device, dtype=torch.float32)
    hip.gemm_h(dyP, xP, 0, 0, N, K, M, C_out=dW, ldc=K)
    return dW


_cb_queued = [False]


_SIDE_MIN_TOKENS = int(os.environ.get('OFB_SIDE_MIN_TOKENS', '12000'))


def _grad_of(t):
    """.grad of a parameter, also when `t` is a fresh reshaping view of it (whose own .grad is always None)"""
    if t.is_leaf:
        return t.grad
    base = t._base
    return base.grad if (base is not None and base.is_leaf) else None


def _side_ok(*params, tokens=None):
    """weight-gradient work may go to the side stream unless a gradient is being accumulated into (AccumulateGrad would add on
    the main stream at once), the feature is off, or the step is host bound anyway (few tokens: the fork / join bookkeeping costs
    ~4 ms of host time per step - DeiT-T bs 8: 13.5 vs 9.4 ms per step - and buys nothing while the GPU waits for launches)"""
    if not hip.SIDE_STREAM or (tokens is not None and tokens < _SIDE_MIN_TOKENS):
        return False
    if any(p is not None and _grad_of(p) is not None for p in params):
        return False
    _end_of_backward_callback()                           # join at the end of this backward pass
    return True


_deferred_params = set()                  # id()s of the parameters that were handed an unreduced (deferred) gradient in this pass


def _end_of_backward_callback():
    """once per backward pass: the side stream is joined and the queued column sums (hip.colsum_deferred) are reduced"""
    if not _cb_queued[0]:
        def _done():
            _cb_queued[0] = False
            _deferred_params.clear()
            hip.join_side()
            hip.flush_deferred()
        torch.autograd.Variable._execution_engine.queue_callback(_done)
        _cb_queued[0] = True


def begin_step():
    """called at the start of every model forward: a backward pass that raised leaves its end-of-pass callback unqueued-but-flagged
    and its deferred jobs behind; start clean"""
    if _cb_queued[0] or _deferred_params or hip._deferred:
        _cb_queued[0] = False
        _deferred_params.clear()
        hip._deferred.clear()


hip._forward_hooks.append(begin_step)


class _nullctx:
    def __enter__(self): return self
    def __exit__(self, *a): return False


def _p_gated_linear_bwd(dyP, dy_colsum, xP, M, W, WP, b, gvec, resid=None, fold=1, ln=None):
    """H-format backward of y = g[n] * (x W^T + b)[n] (or plain Linear when gvec is None): dx (+resid fused), dW, db, dg.
    WP: the forward's H-format copy of W; dy_colsum: callable giving colsum(dY) (the raw bias gradient) as a vector [N] or as
    (partials [rows][N], rows) straight from the producing kernel - the gate-fold kernel adds partial rows up itself."""
    N, K = WP.R, WP.C
    side = _side_ok(W, b, tokens=M)                      # asked of the Parameter itself: a view's .grad is always None
    W = W.view(N, K)
    slot = grad_slot(W)
    if gvec is None:
        dx, _ = p_linear_bwd_input(dyP, M, N, WP, K, resid=resid, ln=ln)
        dW = slot if slot is not None else _new(W, N, K)
        with (hip.side_work(W.device, keep=[dyP.buf, xP.buf]) if side else _nullctx()):
            p_linear_bwd_weight(dyP, xP, M, N, K, out=dW)
        return dx, dW, (_reduced(dy_colsum(), N) if b is not None else None), None
    WeffP = hip.gated_weight_h(W, gvec, N, K)                            # g[n] * W[n][:] as planes (all gated layers in one launch)
    dx, _ = p_linear_bwd_input(dyP, M, N, WeffP, K, resid=resid, ln=ln)
    dbraw, rows = None, 1
    if b is not None:
        dbraw = dy_colsum()
        if isinstance(dbraw, tuple):
            dbraw, rows = dbraw
    dW = slot if slot is not None else _new(W, N, K)
    db, dg, dWraw = (_new(W, N) if b is not None else None), _new(W, N // fold), _new(W, N, K)   # fold: see hip.gate_fold_bwd
    with (hip.side_work(W.device, keep=[dyP.buf, xP.buf, dWraw, dbraw, gvec]) if side else _nullctx()):
        p_linear_bwd_weight(dyP, xP, M, N, K, out=dWraw)
        # (round 6: queueing the 24 folds of a pass for ONE multi-job launch at the end of backward was built and measured - 16.57 /
        #  16.60 / 16.61 against 16.55 / 16.57 / 16.58 ms per step: they already hide on the side stream - and removed again)
        hip.gate_fold_bwd(dWraw, W, gvec, dbraw, b, dW, db, dg, N, K, dbraw_rows=rows, fold=fold)
    return dx, dW, db, dg


def _reduced(db, N):
    """a bias gradient handed over as (partials, rows) summed to a vector"""
    if not isinstance(db, tuple):
        return db
    part, rows = db
    out = _new(part, N)
    hip.colsum(part, N, rows, N, out)
    return out


def bias_grad(dy2d, rowscale=None, rs_div=1):
    M, N = dy2d.shape
    db = _new(dy2d, N)
    hip.colsum(dy2d, N, M, N, db, rowscale=rowscale, rs_div=rs_div)
    return db


def _rs_div(rowscale, M):
    """rowscale holds one factor per token (M entries) or per sample (B entries: every sample's N tokens share it)"""
    return 1 if rowscale is None else M // rowscale.numel()


class Linear(torch.autograd.Function):
    """y = x W^T + b on 2-D inputs (head: vision_transformer.py:744; decoder 1x1 conv :723)."""

    @staticmethod
    def forward(ctx, x2d, W, b):
        x2d, W = _c(x2d), _c(W)
        M, K = x2d.shape
        xP, WP = _P(x2d, M, K), hip.weight_h(W)
        ctx.save_for_backward(xP.buf, W, b)
        ctx.mk, ctx.wp = (M, K), WP
        return p_linear_fwd(xP, M, K, WP, b)[0]

    @staticmethod
    def backward(ctx, dy):
        xbuf, W, b = ctx.saved_tensors
        M, K = ctx.mk
        dy = _c(dy)
        db = _new(dy, W.shape[0]) if b is not None else None
        dyP = hip.to_hformat(dy, M, W.shape[0], W.shape[0], colsum_out=db)
        dx, dW, db, _ = _p_gated_linear_bwd(dyP, lambda: db, _pm(xbuf, M, K), M, W, ctx.wp, b, None)
        return dx, dW.view(W.shape), db


# Hand-overs around LayerNorm on the H-format path (no extra kernels, no extra passes over the activations):
#  * forward: the LN kernel writes its rows as planes too; the wrapper below hangs them on the output tensor (`_ofb_p`), where the
#    branch that consumes it (AttnBranch / MlpBranch -> _P) finds them instead of converting.
#  * backward: the gradient LN returns is the dY of the branch that produced LN's input.  That branch tags its output with its
#    DropPath row scales (`_ofb_up`, see attn_branch / mlp_branch); LN's backward kernel then also writes dx * rowscale as planes
#    and its column sums (the branch's output-bias gradient), parked in _grad_p under the gradient tensor's identity, where the
#    branch's backward picks them up (_take_grad_p) instead of running ofb_to_hformat_colsum over the gradient.
#  * backward, the other way round: the branch that consumes LN's output y computes LN's dy (its input gradient, residual add
#    included) in a GEMM; LN's wrapper tags y with (gamma, rstd) (`_ofb_ln`), the branch hands them to that GEMM, whose epilogue
#    leaves per-tile row-norm maxima (hip.gemm_h(rn=...)), parked in _rn under the gradient's identity: LN's backward kernel takes
#    its output exponent from them instead of running a bound pass over dy (0.34 ms per DeiT-S step).  OFB_RN_HANDOVER=0: off.
_ln_pending = [None]
_LN_F32 = os.environ.get('OFB_LN_F32', '0') == '1'
_grad_p = {}
_rn = {}
_RN_HANDOVER = os.environ.get('OFB_RN_HANDOVER', '1') != '0'


def _put_rn(dx, rn):
    # The entry HOLDS dx: while it is parked nothing can free the gradient and hand its address to another tensor, and autograd
    # cannot accumulate into it in place without moving its version - that is what makes the (data_ptr, version) match in _take_rn
    # sound.  The price is that an entry nobody takes (LayerNorm fell back to its f32 kernel, `y` had a second consumer so autograd
    # summed into a new tensor, an exception mid-backward) would pin an M x D gradient: every LayerNorm backward drops the entry of
    # its dy whether it used it or not, and the tables are emptied at the start of every forward pass (_drop_handovers).
    if len(_rn) >= 4:
        _rn.clear()
    _rn[dx.data_ptr()] = (dx, dx._version, rn)


def _drop_handovers():
    _rn.clear()
    _grad_p.clear()


hip._forward_hooks.append(_drop_handovers)


def _take_rn(dy, rows, D):
    e = _rn.pop(dy.data_ptr(), None)
    if e is None:
        return None
    dx, ver, rn = e
    return rn if (dx._version == ver and dx.numel() == rows * D and dy.is_contiguous()) else None


def _put_grad_p(dx, dxP, colsum, rowscale):
    if len(_grad_p) >= 4:                                    # entries are taken within the same backward pass; drop leftovers
        _grad_p.clear()
    _grad_p[dx.data_ptr()] = (dx, dx._version, dxP, colsum, 0 if rowscale is None else rowscale.data_ptr())


def _take_grad_p(d2, rowscale, M, D):
    """(HMat of d2 * rowscale, its column sums) if LayerNorm's backward already produced them for exactly this tensor"""
    e = _grad_p.pop(d2.data_ptr(), None)
    if e is None:
        return None
    dx, ver, dxP, colsum, rs_ptr = e
    if dx._version != ver or dx.numel() != M * D or dxP.R != M or dxP.C != D or rs_ptr != (0 if rowscale is None else rowscale.data_ptr()):
        return None
    return dxP, colsum


class LayerNorm(torch.autograd.Function):
    """layers.py:96-98 (F.layer_norm, eps inside the sqrt).  Optional `dres` fork: the op also returns its input
    unchanged so that the residual stream's gradient is added inside the backward kernel.  `up`: None, or the 1-tuple
    (rowscale or None) of the branch whose output this LayerNorm reads (see the hand-over note above)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, fork, up):
        x = _c(x)
        D = x.shape[-1]
        rows = x.numel() // D
        y, mean, rstd = torch.empty_like(x), _new(x, rows), _new(x, rows)
        yP = hip.HMat.for_rows_written_by_layernorm(rows, D, x.device)
        # fork (the block LayerNorms: the branch that follows takes the PLANES as its GEMM operand and the LayerNorm's input as its
        # residual, AttnBranch / MlpBranch with `resid` given): nobody reads the f32 rows - they are not written (38.7 MB per call at
        # DeiT-S bs 128, 0.93 GB per step; `y` stays an uninitialised carrier of the planes).  OFB_LN_F32=1 writes them anyway.
        hip.layernorm_fwd_h(x, gamma, beta, y if (not fork or _LN_F32) else None, yP, mean, rstd, rows, D, eps)
        _ln_pending[0] = (yP, (gamma, rstd) if up is not None else None)
        ctx.save_for_backward(x, gamma, mean, rstd)
        ctx.fork = fork
        ctx.up = up
        # deferred parameter-gradient reductions need leaves that are not being accumulated into (see _ln_colsum); the upstream
        # branch's output bias (the third section of the partials) is one of the consumers
        ub = up[1] if (up is not None and len(up) > 1) else None
        ctx.defer_params = (gamma, beta) + ((ub,) if ub is not None else ())
        ctx.defer_ok = all(q.is_leaf and q.grad is None and not q._backward_hooks for q in ctx.defer_params)
        if fork:
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, dy, dres=None):
        x, gamma, mean, rstd = ctx.saved_tensors
        D = x.shape[-1]
        rows = x.numel() // D
        nb = hip.layernorm_bwd_blocks(rows)
        dx = torch.empty_like(x)
        if ctx.up is not None:
            rowscale = ctx.up[0]
            part = _new(x, nb, 3 * D)
            dxP = hip.HMat.for_rows_written_by_layernorm(rows, D, x.device)
            rn = _take_rn(dy, rows, D)                     # (always taken out of the table: see _put_rn)
            if dres is not None:
                rn = None
            if rn is not None:                             # the GEMM that produced dy left the bound's ingredients (see _rn above)
                hip.layernorm_bwd_h_rn(dy, x, gamma, mean, rstd, dx, part, dxP, rowscale, _rs_div(rowscale, rows), rows, D, rn[0], rn[1])
            else:
                hip.layernorm_bwd_h(_c(dy), x, gamma, mean, rstd, _c(dres) if dres is not None else None, dx, part, dxP, rowscale,
                                    _rs_div(rowscale, rows), rows, D)
            dgb = _new(x, 3 * D)
            _ln_colsum(part, 3 * D, nb, dgb, ctx)
            _put_grad_p(dx, dxP, dgb[2 * D:], rowscale)
            return dx, dgb[:D], dgb[D:2 * D], None, None, None
        _rn.pop(dy.data_ptr(), None)                       # a row-norm hand-over parked for this dy has no taker on the f32 path
        part = _new(x, nb, 2 * D)
        hip.layernorm_bwd(_c(dy), x, gamma, mean, rstd, _c(dres) if dres is not None else None, dx, part, rows, D)
        dgb = _new(x, 2 * D)
        _ln_colsum(part, 2 * D, nb, dgb, ctx)
        return dx, dgb[:D], dgb[D:], None, None, None


_DEFER_COLSUM = os.environ.get('OFB_DEFER_COLSUM', '1') != '0'


def _ln_colsum(part, width, nb, dgb, ctx):
    """dgamma | dbeta (| the upstream branch's output-bias gradient) = column sums of LayerNorm's per-block partials.  Their only
    readers are the optimizer and the gradient exchange, so the reduction is queued and runs with all the others of this backward
    pass in ONE launch (hip.flush_deferred) - unless a gradient is being accumulated into right now (AccumulateGrad adds at once)
    or the consumer of the bias part is not a leaf (then ctx.defer_ok is False)."""
    # ... and no parameter may receive a SECOND gradient in this pass while its first one is still unreduced (a LayerNorm module
    # used twice in one graph: autograd would add the two buffers at once)
    ids = [id(q) for q in getattr(ctx, 'defer_params', ())]
    if _DEFER_COLSUM and ctx.defer_ok and not any(i in _deferred_params for i in ids):
        _deferred_params.update(ids)
        hip.colsum_deferred(part, width, nb, width, dgb)
        _end_of_backward_callback()
    else:
        if any(i in _deferred_params for i in ids):
            hip.flush_deferred()                          # the earlier, still unreduced gradient of the shared parameter
        hip.colsum(part, width, nb, width, dgb)


def _ln_apply(x, gamma, beta, eps, fork):
    _ln_pending[0] = None
    out = LayerNorm.apply(x, gamma, beta, eps, fork, getattr(x, '_ofb_up', None))
    pend, _ln_pending[0] = _ln_pending[0], None
    if pend is not None:
        y = out[0] if fork else out
        y._ofb_p = pend[0]
        if pend[1] is not None:
            y._ofb_ln = pend[1]
    return out


def layer_norm(x, gamma, beta, eps):
    return _ln_apply(x, gamma, beta, eps, False)


def layer_norm_fork(x, gamma, beta, eps):
    """returns (LN(x), x): use the second output as the residual input of the following branch."""
    return _ln_apply(x, gamma, beta, eps, True)


def _att_planes_ok(B, N):
    """the plane-writing attention forward lays its 16-token tiles from position -(b N mod 4) and spreads them over as many workgroups
    as they need: every sequence length the kernels take (csrc/attention.hip: ATT_NLIMIT) qualifies"""
    return N <= 4096


class AttnBranch(torch.autograd.Function):
    """out = resid + rowscale[token] * proj(attention(g * qkv(x)))   (layers.py:488-517 + residual/DropPath of
    vision_transformer.py:197,203).  If `resid` is None the branch input is also the residual (the search
    path, where LN output replaces the stream) and its gradient add is fused into the qkv input-grad GEMM.
    g: the gate as the module holds it - (H, d) joint space, (H, 1) head-only, (1, d) channel-only (layers.py:424-448).  The
    broadcast to (H, d) happens HERE and its gradient is reduced back to g's shape inside backward, on the stream that produced
    it: an `expand` outside this Function would put an autograd reduction kernel on the main stream between the side-stream
    gate gradient and the join in BiMaskGates.backward (a race)."""

    @staticmethod
    def forward(ctx, x, resid, wqkv, bqkv, wproj, bproj, g, rowscale, heads, scale, g3_pre=None):
        x = _c(x)
        B, N, D = x.shape
        M = B * N
        x2d = x.view(M, D)
        Hd = wqkv.shape[0] // 3
        dh = Hd // heads
        g3, gshape = None, None
        if g is not None:
            gshape = tuple(g.shape)
            if g.numel() != Hd:
                if g.dim() != 2 or gshape[0] not in (1, heads) or gshape[1] not in (1, dh):
                    raise hip.OfbError(f'attention gate of shape {gshape} does not broadcast to ({heads}, {dh})')
                g = g.expand(heads, dh)
            # the q | k | v tiling of the gate: handed in by the model for all blocks at once (vision_transformer._compute_gates,
            # two launches per forward instead of one per block) or made here
            if g3_pre is not None and g3_pre.numel() == 3 * Hd and g3_pre.is_contiguous() and g.numel() == Hd:
                g3 = g3_pre
            else:
                g3 = g.reshape(-1).repeat(3).contiguous()
        r2d = x2d if resid is None else _c(resid).view(M, D)
        xP, wqP, wpP = _P(x, M, D), hip.weight_h(wqkv), hip.weight_h(wproj)
        ctx.wp = (wqP, wpP)
        ctx.ln = getattr(x, '_ofb_ln', None) if resid is None else None     # (x is the LayerNorm's only consumer then: see _rn)
        qb = _new(x, 1)                                # device-side bound of |qkv| (Cauchy-Schwarz, from the qkv GEMM): the attention
        qkv, _ = p_linear_fwd(xP, M, D, wqP, bqkv, colscale=g3, cbound_out=qb)    # kernels split q, k, v with its exponent
        if g3 is not None:
            hip.gated_register(wqkv, g3, wqkv.shape[0], D)
        o, lse = _new(x, M, Hd), _new(x, 2 * B * heads, N)          # lse travels as two floats (hip.attention_fwd)
        if _att_planes_ok(B, N):                       # the attention kernel writes the projection's operand planes too
            oP = hip.HMat.for_rows_written_by_kernel(M, Hd, x.device)
            hip.attention_fwd_h(qkv, o, oP, lse, B, N, heads, dh, scale, qb)
        else:
            hip.attention_fwd(qkv, o, lse, B, N, heads, dh, scale, qb)
            oP = hip.to_hformat(o, M, Hd, Hd, bound=qb)              # |softmax-weighted mean of v rows| <= max |v|
        out, _ = p_linear_fwd(oP, M, Hd, wpP, bproj, rowscale=rowscale, rs_div=_rs_div(rowscale, M), resid=r2d)
        ctx.save_for_backward(xP.buf, qkv, o, lse, wqkv, bqkv, wproj, g3, rowscale, oP.buf, qb)
        ctx.meta = (B, N, D, heads, dh, scale, resid is None, bproj is not None, gshape)
        return out.view(B, N, D)

    @staticmethod
    def backward(ctx, dout):
        xbuf, qkv, o, lse, wqkv, bqkv, wproj, g3, rowscale, obuf, qb = ctx.saved_tensors
        B, N, D, heads, dh, scale, self_resid, has_pb, gshape = ctx.meta
        M, Hd = B * N, heads * dh
        xP, oP = _pm(xbuf, M, D), _pm(obuf, M, Hd)
        d2 = _c(dout).view(M, D)
        # gradient of the branch output: DropPath factor applied, planes written and the projection's bias gradient summed in ONE pass
        hit = _take_grad_p(d2, rowscale, M, D)
        if hit is not None:
            d2sP, dbp = hit[0], (hit[1] if has_pb else None)
        else:
            dbp = _new(d2, D) if has_pb else None
            d2sP = hip.to_hformat(d2, M, D, D, rowscale=rowscale, rs_div=_rs_div(rowscale, M), colsum_out=dbp)
        wqP, wpP = ctx.wp
        dob = _new(d2, 1)                                       # bound of |dO| for the attention backward's split
        do, _ = p_linear_bwd_input(d2sP, M, D, wpP, Hd, cbound_out=dob)
        dwp = grad_slot(wproj)
        dwp = dwp if dwp is not None else _new(d2, D, Hd)
        with (hip.side_work(d2.device, keep=[d2sP.buf, oP.buf]) if _side_ok(wproj, tokens=M) else _nullctx()):
            p_linear_bwd_weight(d2sP, oP, M, D, Hd, out=dwp)
        # dq | dk | dv leave the attention kernel as f32 rows together with their maxima (one word per workgroup, plainly stored: no
        # atomics, no memset node): the conversion pass reduces them to its exponent and sums the columns (the raw qkv bias
        # gradient) on the way
        dqkv, dq_amax = torch.empty_like(qkv), _new(d2, B * heads)
        hip.attention_bwd(qkv, o, lse, do, dqkv, B, N, heads, dh, scale, qb, dob, wg_amax=dq_amax)
        if bqkv is not None and (3 * Hd) % 16 == 0:
            # the conversion's per-slab column sums go to their consumer as they are: the gate fold (or the plain bias-gradient
            # reduction) adds the partial rows up itself - no column-sum launch per layer
            dqkvP, part, slabs = hip.to_hformat(dqkv, M, 3 * Hd, 3 * Hd, bound=dq_amax, colpart=True)
            dbq_raw = (part, slabs)
        else:
            dbq_raw = _new(d2, 3 * Hd) if bqkv is not None else None
            dqkvP = hip.to_hformat(dqkv, M, 3 * Hd, 3 * Hd, colsum_out=dbq_raw, bound=dq_amax)
        # fold = 3: the gate-fold kernel adds the q | k | v contributions to the gate gradient itself
        dx, dwq, dbq, dg3 = _p_gated_linear_bwd(dqkvP, lambda: dbq_raw, xP, M, wqkv, wqP, bqkv, g3, resid=d2 if self_resid else None, fold=3,
                                                ln=ctx.ln)
        dg = None
        if dg3 is not None:
            dg = dg3.view(heads, dh)
            if gshape != (heads, dh):
                # head-only / channel-only spaces: dg3 may still be in flight on the side stream, so the reduction to the module's gate
                # shape runs there too; the consumer (BiMaskGates.backward) joins the side stream before it reads any gate gradient
                with (hip.side_work(d2.device, keep=[dg3]) if hip._side_dirty[0] else _nullctx()):
                    if gshape[0] == 1:
                        dg = dg.sum(0, keepdim=True)
                    if gshape[1] == 1:
                        dg = dg.sum(1, keepdim=True)
                    if hip._side_dirty[0]:
                        hip._side_keep.append(dg)
        dres = None if self_resid else dout
        return dx.view(B, N, D), dres, dwq, dbq, dwp, dbp, dg, None, None, None, None


# OFB_AUX_T=0: the saved GELU derivative stays row-major and both MLP epilogues park their tiles in LDS (the round-5 path; same-box A/B)
_AUX_T = os.environ.get('OFB_AUX_T', '1') != '0'


class MlpBranch(torch.autograd.Function):
    """out = resid + rowscale[token] * fc2(gelu(g * fc1(x)))   (layers.py:843-865 + residual/DropPath).
    rowscale is the per-sample DropPath factor expanded to one entry per token ([B*N])."""

    @staticmethod
    def forward(ctx, x, resid, w1, b1, w2, b2, g, rowscale):
        x = _c(x)
        B, N, D = x.shape
        M = B * N
        x2d = x.view(M, D)
        gv = None if g is None else _c(g.reshape(-1))
        hid = w1.shape[0]
        # T-layout (the GEMM's own: see below) where every column tile of the hidden width is whole (hid % 192 == 0: all MLP options of
        # DeiT-S / DeiT-B); a ragged last column tile would take the parked epilogue through the layout's index map, slower than row-major
        aux_t = _AUX_T and hid % 192 == 0
        hpre = hip.aux_t(M, hid, x.device) if aux_t else _new(x, M, hid)
        r2d = x2d if resid is None else _c(resid).view(M, D)
        xP, w1P, w2P = _P(x, M, D), hip.weight_h(w1), hip.weight_h(w2)
        ctx.wp = (w1P, w2P)
        ctx.ln = getattr(x, '_ofb_ln', None) if resid is None else None
        # gelu(g * fc1(x)) leaves the kernel as the H-format operand of fc2 (and of the fc2 weight gradient); beside it only
        # GELU'(pre-activation) is kept in f32 (`hpre` holds the derivative here): the epilogue has Phi and phi in hand, and the
        # backward epilogue becomes a single multiply.  It is stored the way the GEMM's waves hold their accumulators (T-layout): both
        # epilogues then run straight from the accumulator registers (gemm_h.hip: the direct epilogue)
        _, hP = p_linear_fwd(xP, M, D, w1P, b1, colscale=gv, act=hip.ACT_GELU_GRAD_T if aux_t else hip.ACT_GELU_GRAD, aux=hpre, want_f32=False,
                             want_p=True)
        if gv is not None:
            hip.gated_register(w1, gv, hid, D)
        out, _ = p_linear_fwd(hP, M, hid, w2P, b2, rowscale=rowscale, rs_div=_rs_div(rowscale, M), resid=r2d)
        ctx.save_for_backward(xP.buf, hpre, hP.buf, w1, b1, w2, gv, rowscale)
        ctx.meta = (B, N, D, resid is None, b2 is not None, aux_t)
        return out.view(B, N, D)

    @staticmethod
    def backward(ctx, dout):
        xbuf, hpre, hbuf, w1, b1, w2, gv, rowscale = ctx.saved_tensors
        B, N, D, self_resid, has_b2, aux_t = ctx.meta
        M, hid = B * N, w1.shape[0]
        xP, hP = _pm(xbuf, M, D), _pm(hbuf, M, hid)
        d2 = _c(dout).view(M, D)
        hit = _take_grad_p(d2, rowscale, M, D)
        if hit is not None:
            d2sP, db2 = hit[0], (hit[1] if has_b2 else None)
        else:
            db2 = _new(d2, D) if has_b2 else None
            d2sP = hip.to_hformat(d2, M, D, D, rowscale=rowscale, rs_div=_rs_div(rowscale, M), colsum_out=db2)
        # d(pre-activation) = (d2s @ W2) * gelu'(hpre): consumed only by the two fc1 gradient products -> H-format only
        w1P, w2P = ctx.wp
        # the fc1 bias gradient (column sums of this H-format-only result) rides on the epilogue
        if b1 is not None:
            _, dhP, part = p_linear_bwd_input(d2sP, M, D, w2P, hid, act=hip.ACT_MULAUX_T if aux_t else hip.ACT_MULAUX, aux=hpre, want_f32=False, want_p=True,
                                              want_colpart=True)
            db1_raw = (part, part.shape[0])                  # per-tile partial sums: added up by their consumer
        else:
            db1_raw = None
            _, dhP = p_linear_bwd_input(d2sP, M, D, w2P, hid, act=hip.ACT_MULAUX_T if aux_t else hip.ACT_MULAUX, aux=hpre, want_f32=False, want_p=True)
        dw2 = grad_slot(w2)
        dw2 = dw2 if dw2 is not None else _new(d2, D, hid)
        with (hip.side_work(d2.device, keep=[d2sP.buf, hP.buf]) if _side_ok(w2, tokens=M) else _nullctx()):
            p_linear_bwd_weight(d2sP, hP, M, D, hid, out=dw2)
        dx, dw1, db1, dg = _p_gated_linear_bwd(dhP, lambda: db1_raw, xP, M, w1, w1P, b1, gv, resid=d2 if self_resid else None, ln=ctx.ln)
        dres = None if self_resid else dout
        return dx.view(B, N, D), dres, dw1, db1, dw2, db2, (None if dg is None else dg.view(1, -1)), None


def attn_branch(x, resid, wqkv, bqkv, wproj, bproj, g, rowscale, heads, scale):
    out = AttnBranch.apply(x, resid, wqkv, bqkv, wproj, bproj, g, rowscale, heads, scale, getattr(g, '_ofb_g3', None))
    out._ofb_up = (rowscale, bproj)                          # for the LayerNorm that reads this output (see LayerNorm)
    return out


def mlp_branch(x, resid, w1, b1, w2, b2, g, rowscale):
    out = MlpBranch.apply(x, resid, w1, b1, w2, b2, g, rowscale)
    out._ofb_up = (rowscale, b2)
    return out


class PatchEmbedTokens(torch.autograd.Function):
    """imgs -> (B, L+1, D) token buffer: conv16/16 as a GEMM over patchified pixels (layers.py:177), embed gate
    (:191), pos-embed, patch masking + mask token, cls row (vision_transformer.py:615-651)."""

    @staticmethod
    def forward(ctx, imgs, wconv, bconv, g, pos, cls, mask_token, mask, patch):
        B, Cin, Hh, Ww = imgs.shape
        gh, gw = Hh // patch, Ww // patch
        L, D = gh * gw, wconv.shape[0]
        # patchify is folded into the operand conversion: the planes of the patch matrix (rows (b, py, px), columns (c, i, j)) are
        # written straight from the images; the backward needs them only as a weight-gradient operand
        w2d = wconv.reshape(D, -1)
        patchesP = hip.patchify_hformat(_c(imgs), patch)
        conv, _ = p_linear_fwd(patchesP, B * L, w2d.shape[1], hip.weight_h(wconv, (D, w2d.shape[1])), bconv)
        patches = patchesP.buf
        tok = _new(imgs, B, L + 1, D)
        gv = None if g is None else _c(g.reshape(-1))
        posv, clsv = _c(pos.reshape(L + 1, D)), _c(cls.reshape(-1))
        mt = None if mask_token is None else _c(mask_token.reshape(-1))
        mk = None if mask is None else _c(mask.reshape(-1))
        hip.embed_assemble_fwd(conv, gv, posv, clsv, mt, mk, tok, B, L, D)
        ctx.save_for_backward(patches, conv, gv, posv, clsv, mt, mk, w2d)
        ctx.meta = (B, L, D, tuple(wconv.shape), tuple(pos.shape), tuple(cls.shape), None if g is None else tuple(g.shape),
                    None if mask_token is None else tuple(mask_token.shape))
        return tok

    @staticmethod
    def backward(ctx, dtok):
        patches, conv, gv, posv, clsv, mt, mk, w2d = ctx.saved_tensors
        B, L, D, wshape, pshape, cshape, gshape, mshape = ctx.meta
        chunks = hip.embed_assemble_chunks(B)
        dconv = torch.empty_like(conv)
        part = _new(conv, 3, chunks, L + 1, D)
        hip.embed_assemble_bwd(_c(dtok), conv, gv, posv, clsv, mt, mk, dconv, part[0], part[1], part[2], B, L, D)
        dpos = _new(conv, L + 1, D)
        hip.splitk_reduce(part[0], chunks, (L + 1) * D, dpos)
        dgm = _new(conv, 2, D)
        hip.colsum(part[1], D, chunks * (L + 1), D, dgm[0])
        hip.colsum(part[2], D, chunks * (L + 1), D, dgm[1])
        Kp = w2d.shape[1]
        db = _new(dconv, D)
        dw = p_linear_bwd_weight(hip.to_hformat(dconv, B * L, D, D, colsum_out=db), _pm(patches, B * L, Kp), B * L, D, Kp, out=grad_slot(w2d))
        # (a copy, not a view of dpos: autograd may adopt a returned gradient as the leaf's .grad; two leaves whose .grad share
        #  storage would both accumulate into it from the second micro-step of a gradient-accumulation window on)
        dcls = dpos[0].reshape(cshape).clone()
        return (None, dw.view(wshape), db, None if gshape is None else dgm[0].view(gshape), dpos.view(pshape), dcls,
                None if mshape is None else dgm[1].view(mshape), None, None)


class TokenTaps(torch.autograd.Function):
    """The two readers of the final token stream (vision_transformer.py:735-744): cls rows [B][D] for the head and the MASKED patch rows
    [n][D] for the decoder.  As two separate autograd ops (a slice and an index_select) the backward built TWO dense zero tensors of the
    stream's size, scattered one gradient into each and added them (5 launches, ~230 MB of traffic at DeiT-S bs 128 for 1408 non-zero
    rows); here ONE zero tensor receives both (cls rows and masked rows are disjoint and unique: plain copies, no accumulation)."""

    @staticmethod
    def forward(ctx, latent, patch_ids):
        """patch_ids: int32 global patch ids b * L + l of the masked patches (ofb_patch_mask's list); their token rows are computed in
        the kernel.  One launch forward (ofb_token_taps_fwd), a memset node + one launch backward (round 6: they were five and three)."""
        B, T, D = latent.shape
        latent = _c(latent)
        n = int(patch_ids.numel())
        cls, z = _new(latent, B, D), _new(latent, n, D)
        hip.token_taps_fwd(latent, patch_ids if n else None, n, B, T, D, cls, z if n else None)
        ctx.save_for_backward(patch_ids)
        ctx.shape = (B, T, D)
        return cls, z

    @staticmethod
    def backward(ctx, dcls, dz):
        (patch_ids,) = ctx.saved_tensors
        B, T, D = ctx.shape
        like = dcls if dcls is not None else dz
        g = torch.empty(B, T, D, device=like.device, dtype=torch.float32)
        n = int(patch_ids.numel())
        hip.token_taps_bwd(None if dcls is None else _c(dcls), None if (dz is None or n == 0) else _c(dz), patch_ids if n else None, n, B, T, D, g)
        return g, None


def norm_targets(imgs, ksize=47):
    """vision_transformer.py:121-141 (no gradient flows through the targets)."""
    imgs = _c(imgs)
    B, Cc, Hh, Ww = imgs.shape
    out, s1, s2 = torch.empty_like(imgs), torch.empty_like(imgs), torch.empty_like(imgs)
    hip.norm_targets(imgs, out, s1, s2, B * Cc, Hh, Ww, ksize)
    return out


def norm_targets_masked(imgs, patch_ids, L, P, ksize=47):
    """norm_targets for the pixels of the listed patches only (ids = b * L + l): the other pixels of the returned buffer are
    uninitialised and must not be read (PmimLoss with the same patch_ids reads exactly the listed patches)."""
    imgs = _c(imgs)
    B, Cc, Hh, Ww = imgs.shape
    out = torch.empty_like(imgs)
    hip.norm_targets_masked(imgs, patch_ids, out, B, Cc, L, P, Hh, Ww, ksize)
    return out


class PmimLoss(torch.autograd.Function):
    """sum(|t - x_rec| * M) / (sum(M) + 1e-5) / C evaluated in patch layout (vision_transformer.py:724-729).
    patch_ids (int32, optional): rec holds only those patches (the masked ones); exact, since M = 0 elsewhere."""

    @staticmethod
    def forward(ctx, rec, targets, mask, patch_ids, B, L, P, Cc):
        rec, mask = _c(rec), _c(mask.reshape(-1))
        n_rows = rec.shape[0]
        partial, out2 = _new(rec, n_rows), _new(rec, 2)
        hip.pmim_loss_fwd(rec, targets, mask, patch_ids, n_rows, partial, out2, B, L, P, Cc)
        ctx.save_for_backward(rec, targets, mask, out2, patch_ids)
        ctx.meta = (B, L, P, Cc, n_rows)
        return out2[0]

    @staticmethod
    def backward(ctx, up):
        rec, targets, mask, out2, patch_ids = ctx.saved_tensors
        B, L, P, Cc, n_rows = ctx.meta
        drec = torch.empty_like(rec)
        hip.pmim_loss_bwd(rec, targets, mask, patch_ids, n_rows, out2, _c(up).reshape(1), drec, B, L, P, Cc)
        return drec, None, None, None, None, None, None, None


class LabelSmoothingCE(torch.autograd.Function):
    """timm LabelSmoothingCrossEntropy (search.py:584): mean_b[(1-s) nll + s mean_c(-logp)]."""

    @staticmethod
    def forward(ctx, logits, labels, smoothing):
        logits = _c(logits)
        Bn, Cn = logits.shape
        row, loss, grad = _new(logits, Bn), _new(logits, 1), torch.empty_like(logits)
        hip.ls_cross_entropy(logits, labels, row, loss, grad, Bn, Cn, smoothing)
        ctx.save_for_backward(grad)
        return loss[0]

    @staticmethod
    def backward(ctx, up):
        (grad,) = ctx.saved_tensors
        out = torch.empty_like(grad)
        hip.scale_by_scalar(grad, _c(up).reshape(1), out, grad.numel())
        return out, None, None


def _scalar_ok(*ts):
    return all(t is None or (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.numel() == 1) for t in ts)


_w4_cache = {}


class ArchLoss(torch.autograd.Function):
    """w1 * attn + w2 * mlp + w4 * embed + w5 * flops (reference losses.py:97-104) from the gate kernel's 3-slot sparsity vector and the
    FLOPs loss, one launch forward (ofb_loss_mix) and one backward (the weights times the upstream scalar)."""

    @staticmethod
    def forward(ctx, spars3, flops, w1, w2, w4, w5):
        out3 = _new(spars3, 3)
        hip.loss_mix(None, spars3, flops, None, (w1, w2, w4, w5), out3)
        key = (float(w1), float(w2), float(w4), float(w5), spars3.device)
        wv = _w4_cache.get(key)
        if wv is None:
            if len(_w4_cache) > 64:
                _w4_cache.clear()
            wv = _w4_cache[key] = torch.tensor(key[:4], device=spars3.device, dtype=torch.float32)
        ctx.wv = wv
        return out3[0]

    @staticmethod
    def backward(ctx, up):
        out = torch.empty_like(ctx.wv)
        hip.scale_by_scalar(ctx.wv, _c(up).reshape(1), out, 4)
        return out[:3], out[3], None, None, None, None


class TotalLoss(torch.autograd.Function):
    """base + arch + stopgrad(base / decoder_loss) * decoder_loss (reference engine.py:134-144), one launch forward; backward: the
    upstream scalar for base and arch, upstream * (base / decoder_loss) for the decoder loss."""

    @staticmethod
    def forward(ctx, base, arch, dec):
        out3 = _new(base, 3)
        hip.loss_mix(base, None, arch, dec, (0.0, 0.0, 0.0, 1.0), out3)
        ctx.coef = out3[1:2]
        ctx.has = (arch is not None, dec is not None)
        return out3[2]

    @staticmethod
    def backward(ctx, up):
        ddec = None
        if ctx.has[1]:
            ddec = torch.empty_like(ctx.coef)
            hip.scale_by_scalar(ctx.coef, _c(up).reshape(1), ddec, 1)
            ddec = ddec[0]
        return up, (up if ctx.has[0] else None), ddec


class FlopsLoss(torch.autograd.Function):
    """((searched - target) / total)^2 over the MAC model of vision_transformer.py:759-783 (base_model.py:31-35).
    n_active: None, or the 0-dim device tensor of the searched model's patch count (vision_transformer.py:768; differentiable
    w.r.t. alpha_patch while more than one patch cell is live)."""

    @staticmethod
    def forward(ctx, wsum, cfg, n_active=None):
        out4, dws = _new(wsum, 4), torch.empty_like(wsum)
        if n_active is not None:
            n_active = n_active.detach().float().contiguous()
            cfg.active_patches = n_active.data_ptr()
        hip.flops_loss(wsum, cfg, out4, dws)
        ctx.save_for_backward(dws, out4)
        ctx.has_n = n_active is not None
        ctx.mark_non_differentiable(out4)
        return out4[0].clone(), out4

    @staticmethod
    def backward(ctx, up, _unused):
        dws, out4 = ctx.saved_tensors
        out = torch.empty_like(dws)
        hip.scale_by_scalar(dws, _c(up).reshape(1), out, dws.numel())
        return out, None, ((out4[3] * up) if ctx.has_n else None)


class BiMaskGates(torch.autograd.Function):
    """All bi-mask gates + adaptive one-hot loss in one launch.  plan: list of dicts (static per module:
    H, C, A0, A1, kind, head_thr, chan_thr, norm_coef) plus per-call `on` (uint8 list) and `w_p`.
    Inputs: alpha_0, score_0, alpha_1, score_1, ...  Outputs: g_0.., wr_0.., wm_0.., wsum[n], spars[3]."""

    @staticmethod
    def forward(ctx, plan, flags, *params):
        n = len(plan)
        dev = params[0].device
        sizes_hc = [p['H'] * p['C'] for p in plan]
        cells = [p['A0'] * p['A1'] for p in plan]
        nblk = [(hc + 255) // 256 for hc in sizes_hc]
        # one flat fp32 buffer for every per-module output, carved below (offsets in floats)
        off, total = [], 0
        for hc, ce, nb in zip(sizes_hc, cells, nblk):
            off.append(total)
            total += 3 * hc + 2 * ce + 2 + nb
        buf = torch.empty(total + n + 3 + n, device=dev, dtype=torch.float32)
        rank = torch.empty(sum(sizes_hc), device=dev, dtype=torch.int32)
        base, rbase = buf.data_ptr(), rank.data_ptr()
        descs = (hip.GateDesc * n)()
        gs, wrs, wms, roff = [], [], [], 0
        wsum_off = total
        for i, p in enumerate(plan):
            a, s = params[2 * i], params[2 * i + 1]
            hc, ce, o = sizes_hc[i], cells[i], off[i]
            d = descs[i]
            d.alpha, d.score = a.data_ptr(), s.data_ptr()
            d.g, d.wr, d.wm = base + 4 * o, base + 4 * (o + hc), base + 4 * (o + 2 * hc)
            d.prob = base + 4 * (o + 3 * hc)
            d.dloss_dalpha = base + 4 * (o + 3 * hc + ce)
            d.loss_alpha = base + 4 * (o + 3 * hc + 2 * ce)
            d.sig_partial = base + 4 * (o + 3 * hc + 2 * ce + 2)
            d.wsum = base + 4 * (wsum_off + i)
            d.rank = rbase + 4 * roff
            d.H, d.C, d.A0, d.A1, d.kind = p['H'], p['C'], p['A0'], p['A1'], p['kind']
            d.w_p, d.norm_coef = p['w_p'], (p['norm_coef'] if flags[2] else 0.0)
            for k, v in enumerate(p['head_thr']):
                d.head_thr[k] = v
            for k, v in enumerate(p['chan_thr']):
                d.chan_thr[k] = v
            for k, v in enumerate(p['on']):
                d.on[k] = v
            shape = (p['H'], p['C'])
            gs.append(buf[o:o + hc].view(shape))
            wrs.append(buf[o + hc:o + 2 * hc].view(shape))
            wms.append(buf[o + 2 * hc:o + 3 * hc].view(shape))
            roff += hc
        descs_dev, host = hip.upload_structs(descs, dev)
        wsum = buf[wsum_off:wsum_off + n]
        spars = buf[wsum_off + n:wsum_off + n + 3]
        per_module = buf[wsum_off + n + 3:wsum_off + 2 * n + 3]
        hip.gates_fwd(descs_dev, n, max(sizes_hc), flags[0], flags[1], flags[2], spars, per_module)
        ctx.keep = (descs_dev, host, buf, rank, plan)
        ctx.shapes = [(params[2 * i].shape, params[2 * i + 1].shape) for i in range(n)]
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(per_module)
        return (*gs, *wrs, *wms, wsum, spars, per_module)

    @staticmethod
    def backward(ctx, *grads):
        hip.join_side()                                   # gate gradients of the gated Linear layers come off the side stream
        descs_dev, _, buf, rank, plan = ctx.keep
        n = len(plan)
        dgs, dwrs, dwms = grads[:n], grads[n:2 * n], grads[2 * n:3 * n]
        dwsum, dspars = grads[3 * n], grads[3 * n + 1]
        dev = buf.device
        dwsum = None if dwsum is None else _c(dwsum)
        dspars = None if dspars is None else _c(dspars)
        gtab = (hip.GateGrad * n)()
        outs, keep = [], []
        for i, p in enumerate(plan):
            ashape, sshape = ctx.shapes[i]
            da = torch.empty(ashape, device=dev, dtype=torch.float32)
            ds = torch.empty(sshape, device=dev, dtype=torch.float32)
            t = gtab[i]
            for name, src in (('dg', dgs[i]), ('dwr', dwrs[i]), ('dwm', dwms[i])):
                if src is not None:
                    src = _c(src)
                    keep.append(src)
                    setattr(t, name, src.data_ptr())
            if dwsum is not None:
                t.dwsum = dwsum.data_ptr() + 4 * i
            if dspars is not None:
                t.dspars = dspars.data_ptr() + 4 * p['kind']
            t.dalpha, t.dscore = da.data_ptr(), ds.data_ptr()
            outs += [da, ds]
        gdev, ghost = hip.upload_structs(gtab, dev)
        hip.gates_bwd(descs_dev, gdev, n)
        ctx.keep2 = (gdev, ghost, keep)
        return (None, None, *outs)
