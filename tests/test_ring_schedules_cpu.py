"""CPU: the slot-reuse arguments of the two LDS rings added in round 4, executed.

* csrc/volume.hip build_volume_walk_kernel: the right window is a ring of NB = WT / DS + 2 blocks of DS pixels (pixel x in slot
  (x - (w0 + 1)) mod (NB * DS)); the loader wave issues the block of step s + 1 at the START of step s (one barrier earlier the compute waves
  finished step s - 1).  Claim: the block it overwrites holds no pixel that step s (or any later step) still reads, and every pixel step s
  reads is in the ring at its slot.
* csrc/conv_kernel.h BL = 1: 4 slots, the transfer of step g + 3 is issued in step g after the barrier, operands of step g + 1 are read in
  step g.  Claim: the slot a transfer overwrites belongs to a step whose MFMAs every wave has issued (it is past the barrier),
  and a step's operands were written and waited for before the barrier that precedes their first read.
"""
import pytest


@pytest.mark.parametrize("WT,DS", [(32, 8), (32, 4), (64, 8), (64, 4), (16, 4)])
@pytest.mark.parametrize("D", [9, 21, 48, 90, 192])
def test_volume_walk_ring_never_overwrites_a_live_pixel(WT, DS, D):
    NB, w0 = WT // DS + 2, 5 * WT
    NR = NB * DS
    slot = lambda x: (x - (w0 + 1)) % NR
    ring = {}                                                   # slot -> pixel it holds
    def load_block(k):                                          # pixels [w0 + 1 + k DS, w0 + (k + 1) DS]
        for x in range(w0 + 1 + k * DS, w0 + 1 + (k + 1) * DS):
            ring[slot(x)] = x
    for k in list(range(WT // DS)) + [-1]:                      # prologue: the initial window
        load_block(k)
    nsteps = (D + DS - 1) // DS
    need = lambda s: {w - d for w in range(w0, w0 + WT) for d in range(s * DS, min((s + 1) * DS, D))}
    for s in range(nsteps):
        if s + 1 < nsteps:                                      # loader, top of step s: block of step s + 1
            k = -(s + 2)
            doomed = {ring.get(slot(x)) for x in range(w0 + 1 + k * DS, w0 + 1 + (k + 1) * DS)} - {None}
            live = set().union(*[need(t) for t in range(s, nsteps)])
            assert not (doomed & live), f"step {s}: the transfer overwrites pixels still needed: {sorted(doomed & live)}"
            load_block(k)
            # ... and the compute waves of step s must not see the NEW pixels where they expect old ones (checked by `doomed & live` above)
        for x in need(s):                                       # compute waves, step s
            assert ring.get(slot(x)) == x, f"step {s}: pixel {x - w0:+d} is not in the ring"


@pytest.mark.parametrize("nsteps", [1, 2, 3, 27, 28, 108])
def test_conv_b_ring_protocol(nsteps):
    SLOTS, AHEAD = 4, 3
    landed, published, holds = {}, set(), {}
    issued_at = {}
    def issue(g, now):
        holds[g % SLOTS] = g
        issued_at[g] = now
    for g in range(AHEAD):                                      # prologue: steps 0..2, waited for before the first barrier
        issue(g, -1)
    waited = set(range(AHEAD))
    last_read = {}
    for g in range(nsteps):
        published |= waited                                     # s_barrier at the top of step g: what every wave waited for is visible
        victim = holds.get((g + AHEAD) % SLOTS)                 # the transfer of step g + 3 overwrites this step's operands
        if victim is not None:
            # its operands were read one step ahead (or at the chunk start) and CONSUMED by the MFMAs of step `victim`: every wave is past
            # that step's MFMAs -- and with them past the LDS reads they waited for -- once it is past the barrier at the top of step g
            assert victim <= g - 1 and last_read.get(victim, -10) <= victim, f"step {g}: slot of step {victim} may still be read"
        issue(g + AHEAD, g)
        if g == 0:
            assert 0 in published
            last_read[0] = 0                                    # the chunk start reads step 0's operands
        assert (g + 1) in published and holds[(g + 1) % SLOTS] == g + 1, f"step {g}: operands of step {g + 1} are not there"
        last_read[g + 1] = g                                    # read one step ahead
        waited |= {t for t in issued_at if issued_at[t] <= g - 1}   # s_waitcnt vmcnt(NIW): everything but this step's own transfer is home
