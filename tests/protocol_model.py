"""Explicit-state models of the persistent kernels' hand-off protocols (test infrastructure, host only; DESIGN.md section
"Hand-off protocols").

A model is a set of PROCESSES (one per workgroup), each a straight-line list of atomic actions over shared COUNTERS and shared MEMORY
words.  `explore` enumerates every interleaving (depth-first over the reachable global states, memoised) and reports the first
    * stale / unpublished read -- a consumer reads a word that does not carry the tag its step expects (a slot of the previous launch,
      of another step of this launch, or a sentinel),
    * overwrite of a word whose reader has not read it yet (shows up as the reader's wrong tag, or as a write onto a non-sentinel word
      in the data-as-flag rings),
    * deadlock -- some process is not finished and no action is enabled.
Actions (tuples):
    ("wait",  [(counter, target), ...])        enabled when every counter >= its target        (one lane polls, barrier: atomic here)
    ("add",   counter)                          counter += 1                                    (the arrival of a publishing workgroup)
    ("write", key, tag)                         mem[key] = tag                                   (a write-through store that has landed)
    ("writes", [keys], tag)                     the same for several words at once (coarser models: fewer states)
    ("write_clean", key, tag)                   the same, but the word must hold SENT            (data-as-flag ring: slot recycled?)
    ("read",  key, tag)                         violation unless mem[key] == tag
    ("reads", [keys], tag)                      several reads at once.  (Sound for finding violations: whatever a single late read can see,
                                                the whole group sees when it is scheduled at that moment.)
    ("resets", [keys])                          several resets at once
    ("poll",  [keys])                           enabled when no word is SENT                     (data-as-flag sweep)
    ("reset", key)                              mem[key] = SENT
Memory order: every action is atomic and takes effect at once -- the model checks the PROTOCOL (what a counter value lets a consumer
infer), not the memory system; the kernels' store -> drain -> arrive and poll -> barrier -> sc1-load orderings are what make an action
atomic in this sense (MI355X_MICROARCH.md, "Valid forms").
"""
SENT = "SENT"
STALE = "stale"          # content of a buffer before the launch (the previous launch's data)


class Violation(Exception):
    pass


def explore(procs, mem0, counters0=None, max_states=4_000_000):
    """procs: list of action lists.  Returns the number of distinct states visited; raises Violation."""
    names = sorted(mem0)
    idx = {k: i for i, k in enumerate(names)}
    cn = sorted({c for p in procs for a in p if a[0] == "add" for c in [a[1]]} |
                {c for p in procs for a in p if a[0] == "wait" for c, _ in a[1]} | set(counters0 or {}))
    cidx = {c: i for i, c in enumerate(cn)}
    # pre-compile
    comp = []
    for p in procs:
        q = []
        for a in p:
            if a[0] == "wait":
                q.append(("wait", tuple((cidx[c], t) for c, t in a[1])))
            elif a[0] == "add":
                q.append(("add", cidx[a[1]]))
            elif a[0] in ("write", "write_clean", "read"):
                q.append((a[0], idx[a[1]], a[2], a[1]))
            elif a[0] == "poll":
                q.append(("poll", tuple(idx[k] for k in a[1])))
            elif a[0] == "writes":
                q.append(("writes", tuple(idx[k] for k in a[1]), a[2]))
            elif a[0] == "reads":
                q.append(("reads", tuple(idx[k] for k in a[1]), a[2], tuple(a[1])))
            elif a[0] == "resets":
                q.append(("resets", tuple(idx[k] for k in a[1])))
            elif a[0] == "reset":
                q.append(("reset", idx[a[1]]))
            else:
                raise ValueError(a)
        comp.append(q)
    n = len(comp)
    lens = [len(q) for q in comp]
    start = (tuple([0] * n), tuple((counters0 or {}).get(c, 0) for c in cn), tuple(mem0[k] for k in names))
    seen = {start}
    stack = [start]
    while stack:
        pcs, ctr, mem = stack.pop()
        any_enabled = False
        for i in range(n):
            pc = pcs[i]
            if pc >= lens[i]:
                continue
            a = comp[i][pc]
            kind = a[0]
            nctr, nmem = ctr, mem
            if kind == "wait":
                if any(ctr[c] < t for c, t in a[1]):
                    continue
            elif kind == "poll":
                if any(mem[k] == SENT for k in a[1]):
                    continue
            elif kind == "add":
                l = list(ctr); l[a[1]] += 1; nctr = tuple(l)
            elif kind == "write":
                l = list(mem); l[a[1]] = a[2]; nmem = tuple(l)
            elif kind == "writes":
                l = list(mem)
                for k in a[1]:
                    l[k] = a[2]
                nmem = tuple(l)
            elif kind == "write_clean":
                if mem[a[1]] != SENT:
                    raise Violation(f"process {i} step {pc}: writes {a[2]} onto {a[3]} = {mem[a[1]]}, a slot its reader has not recycled")
                l = list(mem); l[a[1]] = a[2]; nmem = tuple(l)
            elif kind == "read":
                if mem[a[1]] != a[2]:
                    raise Violation(f"process {i} step {pc}: reads {a[3]} = {mem[a[1]]}, expected {a[2]}")
            elif kind == "reads":
                for k, name in zip(a[1], a[3]):
                    if mem[k] != a[2]:
                        raise Violation(f"process {i} step {pc}: reads {name} = {mem[k]}, expected {a[2]}")
            elif kind == "resets":
                l = list(mem)
                for k in a[1]:
                    l[k] = SENT
                nmem = tuple(l)
            elif kind == "reset":
                l = list(mem); l[a[1]] = SENT; nmem = tuple(l)
            any_enabled = True
            l = list(pcs); l[i] = pc + 1
            st = (tuple(l), nctr, nmem)
            if st not in seen:
                seen.add(st)
                if len(seen) > max_states:
                    raise RuntimeError(f"state space larger than {max_states}")
                stack.append(st)
        if not any_enabled and any(pcs[i] < lens[i] for i in range(n)):
            blocked = {i: procs[i][pcs[i]] for i in range(n) if pcs[i] < lens[i]}
            raise Violation(f"deadlock: {blocked}")
    return len(seen)


# ----------------------------------------------------------------------------------------------------------------------------------
# Encoder backward, lstm_persist_bwd_rs (ast_amd/csrc/lstm_persist.hip: lstm_bwd_rs_steps).  One process per (cell, slice) of ONE batch
# tile; cell = layer (0 = bottom).  Steps s = 0 .. T-1 run time steps t = T-1-s.
#   PR[layer][slot][consumer][producer]   partial dh_rec tiles inside a cell, ring of R slots, slot = t % R
#   PD[layer][t][consumer][producer]      partial dx tiles handed DOWN by `layer` to layer-1, one buffer per time step (no ring)
#   counter A[layer]  own-cell arrivals (counter form of the own-cell hand-off; h = 512)
#   counter B[layer]  arrivals for the down partials of `layer`; the consumers are the slices of layer-1
def encoder_backward_procs(NS, T, layers, sentinel=True, ring=4, up_prefetch=True, last_arrival_fix=True, coarse=False):
    """coarse: a step's loads of one buffer, its product-1 stores, its sentinel resets and its down-partial stores are ONE action each
    (the two-layer models; the one-layer models keep every word's store and reset apart)."""
    tag = lambda t: ("v", t)
    procs, mem, names = [], {}, []

    def reads(keys, tg):
        return [("reads", keys, tg)] if coarse else [("read", k, tg) for k in keys]
    for L in range(layers):
        for slot in range(ring):
            for c in range(NS):
                for p in range(NS):
                    mem[("PR", L, slot, c, p)] = SENT if sentinel else STALE
        if L > 0:
            for t in range(T):
                for c in range(NS):
                    for p in range(NS):
                        mem[("PD", L, t, c, p)] = STALE
    for L in range(layers):
        has_up, has_down = L + 1 < layers, L > 0
        pre = up_prefetch and has_up
        for j in range(NS):
            a = []
            if pre:                                                   # partials of the layer above for the first step, fetched before the loop
                a.append(("wait", [(("B", L + 1), NS)]))
                a += reads([("PD", L + 1, T - 1, j, p) for p in range(NS)], tag(T - 1))
            pending_b = False
            for s in range(T):
                t = T - 1 - s
                if has_up and not pre:                                # (counter form / no prefetch: wait for this step's partials, then load)
                    a.append(("wait", [(("B", L + 1), NS * (s + 1))]))
                    a += reads([("PD", L + 1, t, j, p) for p in range(NS)], tag(t))
                if s > 0:                                             # own cell: the partial tiles of step t+1
                    keys = [("PR", L, (t + 1) % ring, j, p) for p in range(NS)]
                    a.append(("poll", keys) if sentinel else ("wait", [(("A", L), NS * s)]))
                    a += reads(keys, tag(t + 1))
                if pre and t > 0:                                     # in front of the step's barrier: has the layer above published step t-1?
                    a.append(("wait", [(("B", L + 1), NS * (s + 2))]))
                # ---- the step's barrier
                if pending_b:
                    a.append(("add", ("B", L)))                       # down partials of step t+1 (stored a step ago, drained since)
                if pre and t > 0:
                    a += reads([("PD", L + 1, t - 1, j, p) for p in range(NS)], tag(t - 1))
                if coarse:                                            # product 1: this slice's partial tile for every consumer of the cell
                    a.append(("writes", [("PR", L, t % ring, c, j) for c in range(NS)], tag(t)))
                else:
                    for c in range(NS):
                        a.append(("write_clean" if sentinel else "write", ("PR", L, t % ring, c, j), tag(t)))
                if sentinel:
                    if s > 0:
                        rk = [("PR", L, (t + 1) % ring, j, p) for p in range(NS)]
                        a += [("resets", rk)] if coarse else [("reset", k) for k in rk]
                else:
                    a.append(("add", ("A", L)))                       # publish(ctrA): drain, barrier, one arrival
                if has_down:                                          # product 2: partial tiles for the layer below
                    if coarse:
                        a.append(("writes", [("PD", L, t, c, j) for c in range(NS)], tag(t)))
                    else:
                        a += [("write", ("PD", L, t, c, j), tag(t)) for c in range(NS)]
                    pending_b = True
            if pending_b:
                if last_arrival_fix and T > 1:                        # the LAST arrival waits until every peer has made its second-to-last
                    a.append(("wait", [(("B", L), NS * (T - 1))]))
                a.append(("add", ("B", L)))
            procs.append(a)
            names.append((L, j))
    return procs, mem


# ----------------------------------------------------------------------------------------------------------------------------------
# Wide decoder, forward loop (ast_amd/csrc/decoder_wide.hip: decoder_wide_fwd).  Workgroup w runs, per decoder step, the roles it
# owns in program order CELL -> Q -> ATT -> CMB -> CTX.  Phase counters are SHARDED (item i arrives on shard i % NSH; a waiter wants
# ceil((n_items - shard) / NSH) * steps arrivals on every shard); ATT -> CMB uses one counter per batch row.  Buffers are per step
# except PART (the attention partials of a (row, chunk)), which every step overwrites.
def wide_decoder_fwd_procs(W=4, B=2, nsplit=2, q_items=(0,), ctx_items=(1,), cmb0=2, S=3, NSH=2, teacher_forced=True):
    assert B * nsplit <= W and cmb0 + B <= W
    tag = lambda s: ("v", s)
    mem = {}
    for s in range(S + 1):
        for w in range(W):
            mem[("H", s, w)] = STALE                 # h_s of unit group w (HR[s+1] / CVH[s][:, H:])
        for i in range(len(q_items)):
            mem[("Q", s, i)] = STALE
        for b in range(B):
            mem[("CV", s, b)] = STALE
        for i in range(len(ctx_items)):
            mem[("HT", s, i)] = STALE
    for b in range(B):
        for k in range(nsplit):
            mem[("PART", b, k)] = STALE

    def wait_sh(name, n_items, steps):
        return ("wait", [((name, sh), ((n_items - sh + NSH - 1) // NSH) * steps) for sh in range(NSH) if (n_items - sh + NSH - 1) // NSH > 0])
    NQ, NCTX = len(q_items), len(ctx_items)
    procs = []
    for w in range(W):
        a = []
        is_att = w < B * nsplit
        for s in range(S):
            n = s + 1
            # CELL
            if n > 1:
                a.append(wait_sh("cell", W, n - 1))
                a += [("read", ("H", s - 1, u), tag(s - 1)) for u in range(W)]
                a.append(wait_sh("ctx", NCTX, n - 1))
                a += [("read", ("HT", s - 1, i), tag(s - 1)) for i in range(NCTX)]
            a.append(("write", ("H", s, w), tag(s)))
            a.append(("add", ("cell", w % NSH)))
            if w in q_items:
                qi = q_items.index(w)
                a.append(wait_sh("cell", W, n))
                a += [("read", ("H", s, u), tag(s)) for u in range(W)]
                a.append(("write", ("Q", s, qi), tag(s)))
                a.append(("add", ("q", qi % NSH)))
            if is_att:
                b, k = w % B, w // B
                a.append(wait_sh("q", NQ, n))
                a += [("read", ("Q", s, i), tag(s)) for i in range(NQ)]
                a.append(("write", ("PART", b, k), tag(s)))           # overwrites last step's partial
                a.append(("add", ("row", b)))
            if cmb0 <= w < cmb0 + B:
                b = w - cmb0
                a.append(("wait", [(("row", b), nsplit * n)]))
                a += [("read", ("PART", b, k), tag(s)) for k in range(nsplit)]
                a.append(("write", ("CV", s, b), tag(s)))
                a.append(("add", ("cmb", b % NSH)))
            if w in ctx_items:
                ci = ctx_items.index(w)
                if not is_att:
                    a.append(wait_sh("cell", W, n))
                a += [("read", ("H", s, u), tag(s)) for u in range(W)]
                a.append(wait_sh("cmb", B, n))
                a += [("read", ("CV", s, b), tag(s)) for b in range(B)]
                a.append(("write", ("HT", s, ci), tag(s)))
                a.append(("add", ("ctx", ci % NSH)))
        procs.append(a)
    return procs, mem


# Wide decoder, backward loop (decoder_wide_bwd): P1 -> ATTB -> DQC -> P3 -> CELLB -> DZ -> P5R -> next P1.  DHTOP, PARTB, PREC and PCAR
# are single buffers that every step overwrites; DZ items of a tile arrive on per-tile counters (`nq` K-quarters per tile).
def wide_decoder_bwd_procs(W=4, B=2, nsplit=2, p1_items=(0, 1), p3_items=(2,), dqc0=2, p5r_items=(3,), nq=2, S=3, NSH=2):
    assert W % nq == 0 and B * nsplit <= W and dqc0 + B <= W
    NT = W // nq                                    # tiles; workgroup w = tile w // nq, K-part w % nq
    assert len(p5r_items) == NT
    tag = lambda s: ("v", s)
    mem = {}
    for s in range(S + 1):
        for i in range(len(p1_items)):
            mem[("DCVH", s, i)] = STALE
        for b in range(B):
            mem[("DQ", s, b)] = STALE
        for w in range(W):
            mem[("DZ", s, w)] = STALE
        for j in range(NT):
            mem[("DPRE", s, j)] = STALE
    for j in range(NT):
        mem[("DPRE", S - 1, j)] = tag(S - 1)        # d_pre of the last step comes from the batched product in front of the launch
    for i in range(len(p3_items)):
        mem[("DHTOP", i)] = STALE
    for b in range(B):
        for k in range(nsplit):
            mem[("PARTB", b, k)] = STALE
    for w in range(W):
        mem[("PREC", w)] = STALE
        mem[("PCAR", w)] = STALE

    def wait_sh(name, n_items, steps):
        return ("wait", [((name, sh), ((n_items - sh + NSH - 1) // NSH) * steps) for sh in range(NSH) if (n_items - sh + NSH - 1) // NSH > 0])
    NP1, NP3 = len(p1_items), len(p3_items)
    procs = []
    for w in range(W):
        a = []
        tj, kq = w // nq, w % nq
        for st in range(S - 1, -1, -1):
            n = S - st
            if w in p1_items:
                i = p1_items.index(w)
                if n > 1:
                    a.append(wait_sh("dpre", NT, n - 1))
                a += [("read", ("DPRE", st, j), tag(st)) for j in range(NT)]
                a.append(("write", ("DCVH", st, i), tag(st)))
                a.append(("add", ("p1", i % NSH)))
            if w < B * nsplit:
                b, k = w % B, w // B
                a.append(wait_sh("p1", NP1, n))
                a += [("read", ("DCVH", st, i), tag(st)) for i in range(NP1)]
                a.append(("write", ("PARTB", b, k), tag(st)))
                a.append(("add", ("row", b)))
            if dqc0 <= w < dqc0 + B:
                b = w - dqc0
                a.append(("wait", [(("row", b), nsplit * n)]))
                a += [("read", ("PARTB", b, k), tag(st)) for k in range(nsplit)]
                a.append(("write", ("DQ", st, b), tag(st)))
                a.append(("add", ("dqc", b % NSH)))
            if w in p3_items:
                i = p3_items.index(w)
                a.append(wait_sh("dqc", B, n))
                a += [("read", ("DQ", st, b), tag(st)) for b in range(B)]
                a.append(("write", ("DHTOP", i), tag(st)))
                a.append(("add", ("p3", i % NSH)))
            # CELLB (every workgroup)
            a.append(wait_sh("p3", NP3, n))
            if n > 1:
                a.append(("wait", [(("rec", tj), nq * (n - 1))]))
            a += [("read", ("DHTOP", i), tag(st)) for i in range(NP3)]
            if n > 1:
                a += [("read", ("PREC", tj * nq + k), tag(st + 1)) for k in range(nq)]
            a.append(("write", ("DZ", st, w), tag(st)))
            a.append(("add", ("cellb", w % NSH)))
            if st == 0:
                break
            # DZ (every workgroup)
            a.append(wait_sh("cellb", W, n))
            a += [("read", ("DZ", st, u), tag(st)) for u in range(W)]
            a.append(("write", ("PCAR", w), tag(st)))
            a.append(("add", ("car", tj)))
            a.append(("write", ("PREC", w), tag(st)))
            a.append(("add", ("rec", tj)))
            if w in p5r_items:
                j = p5r_items.index(w)
                a.append(("wait", [(("car", j), nq * n)]))
                a += [("read", ("PCAR", j * nq + k), tag(st)) for k in range(nq)]
                a.append(("write", ("DPRE", st - 1, j), tag(st - 1)))
                a.append(("add", ("dpre", j % NSH)))
        procs.append(a)
    return procs, mem


# ----------------------------------------------------------------------------------------------------------------------------------
# Split tiles of the stream-K GEMMs (ast_amd/csrc/gemm.hip, "Split tiles without a zeroing launch").  One process per contributing
# workgroup (per multiplying wave, really: every wave runs the protocol on its own word); the processes BRANCH on what their atomics
# return, so this model has its own little explorer.  Word fields: arrived k-iterations, DONE, departed k-iterations.
#   step 0  old = fetch_add(word, nk)                            first = (old word == 0)
#   first:  step 1 store the sub-tile     step 2 fetch_add(word, DONE | nk << 16)
#   others: step 1 wait for DONE          step 2 add onto the sub-tile      step 3 fetch_add(word, nk << 16)
#   the departure that completes kt iterations writes the word back to zero; a wave whose ARRIVAL already returned "DONE up, everybody
#   before me departed, my iterations complete the tile" skips step 3 and writes the zero itself (one round trip less on the usual path)
# Violations: a contribution added onto a tile nobody has stored yet, a store onto a tile that already holds data, a wave left waiting
# for a DONE that can no longer come (deadlock), a word that is not zero -- or a tile that is not the sum of all contributions -- at the end.
def explore_gemm_ticket(nks, launches=2, reset_by="departure", done_before_store=False, max_states=2_000_000):
    """nks: k-iterations of the contributors of ONE split tile.  launches: the same word serves this many launches one after the other
    (stream order: a launch starts when the one before has finished).  reset_by="arrival": the (wrong) variant in which the contributor
    whose ARRIVAL completes the tile resets the word behind its own add.  done_before_store: the (wrong) variant in which the first
    arrival raises DONE in front of its stores.  Returns the number of states visited; raises Violation."""
    n, kt = len(nks), sum(nks)
    visited = 0
    for launch in range(launches):
        # state: (pcs, firsts, olds, word=(arrived, done, departed), tile=(stored, adds))
        start = (tuple([0] * n), tuple([False] * n), tuple([(0, False)] * n), (0, False, 0), (False, 0))
        seen, stack = {start}, [start]
        while stack:
            pcs, firsts, olds, word, tile = stack.pop()
            arrived, done, departed = word
            stored, adds = tile
            enabled = False
            finished = all(pc == 99 for pc in pcs)
            if finished:
                if word != (0, False, 0):
                    raise Violation(f"launch {launch}: ticket word left as {word}")
                if not stored or adds != n - 1:
                    raise Violation(f"launch {launch}: tile holds {tile}, not one store and {n - 1} adds")
                continue
            for i in range(n):
                pc, nk = pcs[i], nks[i]
                if pc == 99:
                    continue
                nword, ntile, nfirst, nold, npc = word, tile, firsts[i], olds[i], None
                if pc == 0:                                   # arrival
                    nfirst = word == (0, False, 0)
                    # (old arrivals, and whether this arrival already shows that the wave is the last to leave: DONE up, everybody before
                    #  it departed, its own iterations complete the tile -- it then skips poll and departure and resets with a plain store)
                    nold = (arrived, done and departed == arrived and arrived + nk == kt)
                    nword = (arrived + nk, done, departed)
                    npc = 1
                elif firsts[i]:
                    store_pc, done_pc = (2, 1) if done_before_store else (1, 2)
                    if pc == store_pc:
                        if stored or adds:
                            raise Violation(f"launch {launch}: contributor {i} stores onto a tile that holds data {tile}")
                        ntile = (True, adds)
                        npc = 99 if done_before_store else 2
                    elif pc == done_pc:                        # DONE goes up with the first arrival's departure
                        nword = (arrived, True, departed + nk)
                        if departed + nk == kt:
                            nword = (0, False, 0)
                        npc = 2 if done_before_store else 99
                else:
                    if pc == 1:                                # poll
                        if not done:
                            continue
                        npc = 2
                    elif pc == 2:                              # atomic adds
                        if not stored:
                            raise Violation(f"launch {launch}: contributor {i} adds onto a tile nobody has stored")
                        ntile = (stored, adds + 1)
                        if reset_by == "arrival":
                            if olds[i][0] + nk == kt:
                                nword = (0, False, 0)
                            npc = 99
                        elif olds[i][1]:                       # known last at its arrival: plain store of zero, no departure
                            nword = (0, False, 0)
                            npc = 99
                        else:
                            npc = 3
                    elif pc == 3:                              # departure
                        nword = (arrived, done, departed + nk)
                        if departed + nk == kt:
                            nword = (0, False, 0)
                        npc = 99
                enabled = True
                lp, lf, lo = list(pcs), list(firsts), list(olds)
                lp[i], lf[i], lo[i] = npc, nfirst, nold
                st = (tuple(lp), tuple(lf), tuple(lo), nword, ntile)
                if st not in seen:
                    seen.add(st)
                    if len(seen) > max_states:
                        raise RuntimeError(f"state space larger than {max_states}")
                    stack.append(st)
            if not enabled:
                raise Violation(f"launch {launch}: deadlock, contributors at steps {pcs} with word {word}")
        visited += len(seen)
    return visited


# ----------------------------------------------------------------------------------------------------------------------------------
# Round 6: work beside the recurrences.
# (a) Forward: the layer-0 input projection arrives in time chunks from the side stream; chunk k = steps [s0 + k cs, ...) is in memory when
#     flag[k] is up (a one-lane kernel BEHIND the chunk's product: the kernel boundary is the release).  The layer-0 cells (one process
#     per workgroup) enter a chunk behind its flag.  flag_first=True models the bug of raising the flag in front of the product.
def side_chunk_flag_procs(T=6, s0=2, cs=2, nwg=2, flag_first=False):
    tag = lambda t: ("zx", t)
    mem = {("ZX", t, w): (tag(t) if t < s0 else STALE) for t in range(T) for w in range(nwg)}
    nch = (T - s0 + cs - 1) // cs
    prod = []
    for k in range(nch):
        steps = range(s0 + k * cs, min(T, s0 + (k + 1) * cs))
        wr = [("writes", [("ZX", t, w) for t in steps for w in range(nwg)], None)]      # placeholder, split per step below (tags differ)
        wr = [("writes", [("ZX", t, w) for w in range(nwg)], tag(t)) for t in steps]
        prod += ([("add", ("flag", k))] + wr) if flag_first else (wr + [("add", ("flag", k))])
    procs = [prod]
    for w in range(nwg):
        a = []
        for t in range(T):
            if t >= s0 and (t - s0) % cs == 0:
                a.append(("wait", [(("flag", (t - s0) // cs), 1)]))
            a.append(("read", ("ZX", t, w), tag(t)))
        procs.append(a)
    return procs, mem


# (b) Backward: every workgroup of a layer-0 cell writes dz through and arrives on the progress counter once per chunk of cs loop steps --
#     the arrival of a chunk that ends with step s is DEFERRED to the barrier of step s + 1 (its stores have drained by then), the last chunk's
#     follows the loop.  The workgroups of a cell are coupled by the recurrence itself (step s needs every peer's partials of step s - 1:
#     counter "A" here).  A consumer (the side stream's wait kernel + the products behind it) takes chunk k when the counter shows
#     NS (k + 1) arrivals.  last_arrival_fix: the arrival behind the loop first waits for NS (chunks - 1) -- without it an early finisher's
#     last arrival completes the count for chunk k - 1 while a peer has not stored that chunk's last dz (the round-4 race's shape).
def progress_counter_procs(NS=2, T=5, cs=2, last_arrival_fix=True):
    tag = lambda s: ("dz", s)
    mem = {("DZ", s, j): STALE for s in range(T) for j in range(NS)}
    nch = (T + cs - 1) // cs
    procs = []
    for j in range(NS):
        a, pending = [], False
        for s in range(T):
            if s > 0:
                a.append(("wait", [("A", NS * s)]))               # the peers' partials of step s - 1
            if pending:                                            # behind the step's barrier: the deferred arrival of the chunk that ended with s - 1
                a.append(("add", "P"))
                pending = False
            a.append(("add", "A"))                                 # product-1 partials of step s (what the peers' next step waits for)
            a.append(("write", ("DZ", s, j), tag(s)))              # dz of step s: behind the product-1 stores in program order
            if (s + 1) % cs == 0 and s < T - 1:
                pending = True
        if last_arrival_fix and nch > 1:
            a.append(("wait", [("P", NS * (nch - 1))]))
        a.append(("add", "P"))
        procs.append(a)
    cons = []
    for k in range(nch):
        cons.append(("wait", [("P", NS * (k + 1))]))
        for s in range(k * cs, min(T, (k + 1) * cs)):
            cons += [("read", ("DZ", s, j), tag(s)) for j in range(NS)]
    procs.append(cons)
    return procs, mem
