"""Model check of the persistent kernels' hand-off protocols (round-4 review, item 4): every interleaving of small instances of
  * the encoder backward's own-cell hand-off (data-as-flag ring / counter A) and its down-partials counter B, two layers,
  * the h = 512 form of the same kernel (counters on both hand-offs, no early fetch),
  * the wide decoder's phase counters, forward and backward loop,
  * the ticket words of the stream-K GEMMs' split tiles (round 5: first arrival stores, the others add behind its DONE bit),
is enumerated; no consumer may read a slot that does not carry its step's data, no slot may be overwritten before its reader is done
with it, nobody may dead-lock.  The round-4 race (the LAST arrival on counter B needs nothing from the peers) is found by the model
when the fix is taken out of it.  DESIGN.md, section "Hand-off protocols", states the invariants these models encode; the kernels are
ast_amd/csrc/lstm_persist.hip (lstm_bwd_rs_steps) and ast_amd/csrc/decoder_wide.hip."""
import pytest

import protocol_model as M
from protocol_model import Violation, encoder_backward_procs, explore, explore_gemm_ticket, wide_decoder_bwd_procs, wide_decoder_fwd_procs


def test_own_cell_sentinel_ring_needs_three_slots():
    """Data-as-flag ring of the partial tiles inside a cell (h <= 256): a producer may write slot t % R only when the consumer has put
    the sentinel back behind its read of step t + R.  The consumer's reset of slot (t+1) sits behind its own product-1 stores of step t
    in program order, so a producer that has consumed those stores can be one more step ahead: R = 2 is too few, 3 suffice, the kernel
    uses 4."""
    for ring in (4, 3):
        procs, mem = encoder_backward_procs(NS=3, T=5, layers=1, sentinel=True, ring=ring)
        assert explore(procs, mem) > 100
    procs, mem = encoder_backward_procs(NS=3, T=5, layers=1, sentinel=True, ring=2)
    with pytest.raises(Violation):
        explore(procs, mem)


def test_own_cell_counter_a_is_self_limiting():
    """Counter form of the same hand-off (h = 512): EVERY arrival on counter A is preceded by a wait on counter A (count >= NS * s before
    step s), which keeps the slices at most one arrival apart -- "count >= NS * s" then does mean "every slice has published s steps" --
    and with it two ring slots are enough."""
    for ring in (4, 2):
        procs, mem = encoder_backward_procs(NS=3, T=5, layers=1, sentinel=False, ring=ring)
        assert explore(procs, mem) > 100
    procs, mem = encoder_backward_procs(NS=3, T=5, layers=1, sentinel=False, ring=1)
    with pytest.raises(Violation):
        explore(procs, mem)


@pytest.mark.parametrize("form", ["sentinel ring + early fetch (h <= 256)", "counters on both hand-offs (h = 512)"])
def test_counter_b_last_arrival_must_wait_for_the_slowest_peer(form):
    """Two layers, NS = 3 slices, T = 4 steps.  The layer below reads "count_B >= NS * k" as "every slice of the layer above has stored
    its down partials of k steps".  Arrival k < T of a slice happens behind its barrier of step k, which it reaches only behind its
    peers' product-1 stores of step k-1, i.e. behind their arrival k-1: the slices are at most one arrival apart and the inference is
    sound.  The T-th arrival follows the loop and needs nothing from the peers: without the wait `count_B >= NS * (T-1)` in front of it
    an early finisher lifts the count to NS * (T-1) while a slow peer has made only T-2 arrivals, and the layer below reads that peer's
    tile of step 1 before it is written (the previous launch's tile).  The model finds exactly that when the wait is removed."""
    kw = dict(sentinel=True, up_prefetch=True) if form.startswith("sentinel") else dict(sentinel=False, up_prefetch=False)
    procs, mem = encoder_backward_procs(NS=3, T=4, layers=2, last_arrival_fix=True, coarse=True, **kw)
    assert explore(procs, mem) > 1000
    procs, mem = encoder_backward_procs(NS=3, T=4, layers=2, last_arrival_fix=False, coarse=True, **kw)
    with pytest.raises(Violation, match="stale"):
        explore(procs, mem)


def test_wide_decoder_forward_phase_counters():
    """CELL -> Q -> ATT -> CMB -> CTX -> next CELL on sharded phase counters and per-row counters: every consumer finds its step's data,
    the one buffer that is reused from step to step (the attention partials) is not overwritten before the combine has read it, and the
    fixed role order inside a workgroup cannot dead-lock.  (What protects the partials is program order: ATT of step s+1 needs Q of step
    s+1, which needs EVERY workgroup's CELL of step s+1, which every workgroup runs behind all its roles of step s.)  A combine that
    waits for one arrival less than its row's chunks make reads a partial that is not there yet: the checker sees it."""
    procs, mem = wide_decoder_fwd_procs(W=4, B=2, nsplit=2, q_items=(0,), ctx_items=(1,), cmb0=2, S=3, NSH=2)
    assert explore(procs, mem) > 1000

    def short(a):
        if a[0] == "wait" and a[1][0][0][0] == "row":
            return ("wait", [(c, t - 1) for c, t in a[1]])
        return a
    with pytest.raises(Violation):
        explore([[short(a) for a in p] for p in procs], mem)


def test_wide_decoder_backward_phase_counters():
    """P1 -> ATTB -> DQC -> P3 -> CELLB -> DZ -> P5R -> next P1; DHTOP, PARTB, PREC and PCAR are single buffers overwritten every step,
    the K-part partials of a tile arrive on per-tile counters.  A per-tile wait that is one arrival short (the kind of off-by-one the
    round-4 race was) is caught."""
    procs, mem = wide_decoder_bwd_procs(W=4, B=2, nsplit=2, p1_items=(0, 1), p3_items=(2,), dqc0=2, p5r_items=(2, 3), nq=2, S=3, NSH=2)
    assert explore(procs, mem) > 1000

    def short(a):
        if a[0] == "wait" and a[1][0][0][0] == "car":
            return ("wait", [(c, t - 1) for c, t in a[1]])
        return a
    with pytest.raises(Violation):
        explore([[short(a) for a in p] for p in procs], mem)


@pytest.mark.parametrize("nks", [(3, 5), (2, 2, 4), (1, 6, 2, 3), (4, 1, 1, 1, 1)])
def test_gemm_split_tile_ticket_word(nks):
    """Split tiles of the stream-K GEMMs without a zeroing launch (gemm.hip): whoever arrives first -- in every interleaving of 2 to 5
    contributors -- stores, nobody adds before those stores are out, nobody is left polling, and the word is back at zero for the next
    launch.  Two wrong variants must be caught: (a) DONE raised in front of the stores, (b) the word reset by the contributor whose
    ARRIVAL completes the tile (a slower contributor that has arrived but not yet seen DONE then polls a word that has been cleared)."""
    assert explore_gemm_ticket(nks) > 2 * len(nks)
    with pytest.raises(Violation, match="nobody has stored"):
        explore_gemm_ticket(nks, done_before_store=True)
    if len(nks) >= 3:
        with pytest.raises(Violation, match="deadlock|ticket word"):
            explore_gemm_ticket(nks, reset_by="arrival")


def test_side_stream_chunk_flags_and_progress_counter():
    """Round 6 (work beside the recurrences).  Forward: a chunk flag raised BEHIND the chunk's product lets the layer-0 cells read only finished
    rows; raised in front of it, some interleaving reads a row of the previous launch.  Backward: with the deferred per-chunk arrivals and the
    last arrival's wait, `count >= workgroups x (k + 1)` means every workgroup has stored chunk k's dz; WITHOUT that wait an early finisher
    completes the count for the second-to-last chunk while a peer has not stored its last dz -- the checker finds the stale read (the shape
    of round 4's counter-B race, caught here before it ever ran)."""
    procs, mem = M.side_chunk_flag_procs(T=6, s0=2, cs=2, nwg=2)
    assert M.explore(procs, mem) > 0
    procs, mem = M.side_chunk_flag_procs(T=6, s0=2, cs=2, nwg=2, flag_first=True)
    with pytest.raises(M.Violation):
        M.explore(procs, mem)
    for T, cs in ((5, 2), (4, 2), (7, 3), (4, 1)):
        procs, mem = M.progress_counter_procs(NS=2, T=T, cs=cs)
        assert M.explore(procs, mem) > 0
    procs, mem = M.progress_counter_procs(NS=2, T=5, cs=2, last_arrival_fix=False)
    with pytest.raises(M.Violation):
        M.explore(procs, mem)
