"""Static hazard check of a plan's two-stream schedule (the second stream's safety rested on buffer parking matched by address range
and on tests that observed no bug; this walks the op records and PROVES the property, so a violation fails plan compilation on any host).

Executor contract (csrc/api.hip: pdae_run_ops): an op flagged PDAE_OPF_SIDE starts when everything in front of it in the array has finished on
the caller's stream and runs on the library's second stream, in order with the other side ops; PDAE_OP_JOIN (and the end of every
pdae_run_ops call) makes the caller's stream wait for all of them.  Hence the only unordered pairs are (side op S, main-stream op M) with
S in front of M and no join in between.  For every such pair this check requires
    writes(M) ∩ (reads(S) ∪ writes(S)) = ∅   and   reads(M) ∩ writes(S) = ∅
at the granularity of whole pool buffers (a pointer is mapped to the pool allocation that contains it; persistent tensors -- parameters,
gradients, statistics partials, prepared weights -- by their exact address, flat-gradient views are distinct addresses).  Workspaces: main
ops use Plan.ws, side ops Plan.ws_side (two allocations; patched at compile), so a workspace slot never conflicts across the streams.
Ops whose operands live in a device-side table (grouped linears) carry them as a Python attribute `rw = (reads, writes)` set by the builder.
`Plan.run(first, last)` segments (the DDP buckets of trainer/fused_step.py) only ADD joins (every call joins at its end): a schedule that is
safe as a whole is safe in segments.
"""
import bisect

from . import hip as H

# pointer slots an op WRITES (read-modify-write outputs included); every other non-null pointer slot is read.  "ws" slots are listed too: they
# are excluded by identity with the plan's workspaces below.
_W = {
    "OP_CONV_FWD": (5, 19), "OP_CONV_FWD_GN": (6, 19), "OP_CONV_FWD_SKIP": (9, 19), "OP_CONV_SKIP_WPREP": (1,), "OP_CONV_DGRAD": (2, 8),
    "OP_AMAX": (1,), "OP_CONV_WPREP": (1,), "OP_CONV_WGRAD": (3, 4, 5), "OP_GEMM": (2,), "OP_ATTN_FWD": (1, 2), "OP_ATTN_BWD": (4, 5),
    "OP_GN_STATS": (2, 3, 4), "OP_GN_STATS_COEF": (6, 7, 8, 9, 10), "OP_GN_COEF_FROM_CONV_STATS": (6, 7, 8), "OP_GN_STATS_QUADS": (1,),
    "OP_GN_COEF": (6,), "OP_GN_APPLY": (3, 4), "OP_GN_BWD": (10, 11, 12, 13, 14, 15, 16, 17, 18), "OP_MLP_MODLN_FWD": (4, 5, 6),
    "OP_MLP_MODLN_BWD": (7, 8, 9, 10), "OP_TEMB": (2,), "OP_SILU": (1,), "OP_SUBSAMPLE2": (1,), "OP_ZERO_INSERT2": (1,), "OP_SILU_BWD": (2,),
    "OP_AXPBY": (1,), "OP_EMBEDDING": (2,), "OP_EMBEDDING_BWD": (2,), "OP_TO_NHWC": (1,), "OP_FROM_NHWC": (1,), "OP_Q_SAMPLE": (5,),
    "OP_LOSS": (6, 7, 8, 9), "OP_DDIM_STEP": (3,), "OP_DDPM_STEP": (4,), "OP_AXPBY_ROWS": (4,), "OP_DDIM_STEP_ROWS": (4,),
    "OP_DDPM_STEP_ROWS": (6,), "OP_ADAM_EMA": (0, 2, 3, 4, 5), "OP_SOFTMAX": (0,), "OP_SOFTMAX_BWD": (1,), "OP_COLSUM": (1, 2), "OP_MEMSET": (0,),
    "OP_COPY": (1,),
}
_TABLE_OPS = ("OP_LINEAR_GROUP", "OP_LINEAR_BWD_GROUP", "OP_CONV_WPREP_GROUP")      # operands in a device table: need op.rw
WRITES = {getattr(H, k): frozenset(v) for k, v in _W.items() if hasattr(H, k)}
TABLE_KINDS = frozenset(getattr(H, k) for k in _TABLE_OPS if hasattr(H, k))
NPTR = len(H.PdaeOp().p)


class PlanHazard(AssertionError):
    pass


def _unit(spans, a):
    """Pool allocation containing address a (its start), or a itself for a tensor outside the pool."""
    j = bisect.bisect_right(spans, (a, 1 << 62)) - 1
    if j >= 0 and spans[j][0] <= a < spans[j][1]:
        return spans[j][0]
    return a


def op_sets(op, spans, skip=()):
    """(reads, writes) of one record as sets of units.  `skip`: addresses to ignore (the plan's two workspaces)."""
    kind = op.kind
    if kind in TABLE_KINDS:
        rw = getattr(op, "rw", None)
        if rw is None:
            raise PlanHazard(f"op kind {kind} keeps its operands in a device table and carries no rw attribute: the schedule cannot be checked")
        conv = lambda ts: {_unit(spans, t if isinstance(t, int) else t.data_ptr()) for t in ts if t is not None}
        return conv(rw[0]), conv(rw[1])
    if kind not in WRITES and kind != H.OP_JOIN:
        raise PlanHazard(f"op kind {kind} has no entry in plancheck.WRITES")
    wslots = WRITES.get(kind, frozenset())
    r, w = set(), set()
    for k in range(NPTR):
        a = op.p[k]
        if not a or a in skip:
            continue
        (w if k in wslots else r).add(_unit(spans, a))
    return r, w


def check_plan(plan, recs=None):
    """Raises PlanHazard when a main-stream op between a side op and its join touches what the side op writes, or writes what it reads.
    Returns {"side_ops", "joins", "pairs"}: how much was checked."""
    recs = plan.recs if recs is None else recs
    spans = plan._spans
    skip = {t.data_ptr() for t in (getattr(plan, "ws", None), getattr(plan, "ws_side", None)) if t is not None}
    pend_r, pend_w, pend_idx = {}, {}, []          # unit -> index of the pending side op that reads / writes it
    n_side = n_join = pairs = 0
    for k, op in enumerate(recs):
        if op.kind == H.OP_JOIN:
            pend_r.clear(); pend_w.clear(); pend_idx.clear()
            n_join += 1
            continue
        r, w = op_sets(op, spans, skip)
        if op.flags & H.OPF_SIDE:
            n_side += 1
            pend_idx.append(k)
            for u in r:
                pend_r.setdefault(u, k)
            for u in w:
                pend_w.setdefault(u, k)
            continue
        if not pend_idx:
            continue
        pairs += len(pend_idx)
        for u in w:
            if u in pend_r or u in pend_w:
                s = pend_r.get(u, pend_w.get(u))
                raise PlanHazard(f"op {k} (kind {op.kind}, main stream) writes buffer 0x{u:x} that side op {s} (kind {recs[s].kind}) "
                                 f"{'reads' if u in pend_r else 'writes'} with no join in between")
        for u in r:
            if u in pend_w:
                s = pend_w[u]
                raise PlanHazard(f"op {k} (kind {op.kind}, main stream) reads buffer 0x{u:x} that side op {s} (kind {recs[s].kind}) writes with no join in between")
    return {"side_ops": n_side, "joins": n_join, "pairs": pairs}
