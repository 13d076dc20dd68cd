"""CPU: the multi-rank schedule of the native RCCL gather pipeline (vpt_gather_plan — the pure function gather_enqueue_frame
executes, vpt_amd/csrc/vpt_hip.hip), for world sizes a one-GPU box cannot run.

Checked for world 2..8, every root and the all_gather mode, over more than two turns of the buffer ring:
  * every rank derives the same ring buffer / half / event edges for a frame;
  * rooted: each non-root rank sends exactly once to the root; the root posts one receive per peer into disjoint slots
    that, together with its own in-place render slot, tile the receive buffer exactly; nobody else receives;
  * all_gather: every rank renders into its send buffer and takes part in one collective;
  * ring safety under the two-stream event protocol: when the kernel of frame f overwrites ring buffer b, the exchange of the
    frame that used b one ring earlier has been waited for (simulated with per-stream logical clocks: a hipStreamWaitEvent
    observes the latest hipEventRecord enqueued before it)."""
import ctypes as C

import pytest

from vpt_amd import _native as N

OP_NONE, OP_ALLGATHER, OP_SEND, OP_RECV = 0, 1, 2, 3


class Step(C.Structure):
    _fields_ = [("ring", C.c_int), ("buffer", C.c_int), ("parity", C.c_int), ("wait_gathered", C.c_int), ("record_gathered", C.c_int),
                ("rendered_event", C.c_int), ("in_place", C.c_int), ("op", C.c_int), ("peer", C.c_int), ("npeers", C.c_int),
                ("render_offset", C.c_uint64)]


def plan(frame, rank, world, root, nbytes):
    L = N.lib()
    L.vpt_gather_plan.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_uint64, C.POINTER(Step)]
    L.vpt_gather_plan_recv.argtypes = [C.POINTER(Step), C.c_int, C.c_int, C.c_uint64, C.POINTER(C.c_int), C.POINTER(C.c_uint64)]
    st = Step()
    N.check(L.vpt_gather_plan(frame, rank, world, root, nbytes, C.byref(st)))
    recvs = []
    for i in range(st.npeers if st.op == OP_RECV else 0):
        p, off = C.c_int(), C.c_uint64()
        N.check(L.vpt_gather_plan_recv(C.byref(st), rank, i, nbytes, C.byref(p), C.byref(off)))
        recvs.append((p.value, off.value))
    return st, recvs


@pytest.mark.parametrize("world", [1, 2, 3, 4, 5, 8])
def test_schedule_is_consistent_across_ranks(world):
    nbytes = 1920 * 136 * 8
    roots = [-1, 0] + ([world - 1] if world > 1 else []) + ([1] if world > 2 else [])
    for root in roots:
        ring = None
        for frame in range(0, 40):
            steps = [plan(frame, r, world, root, nbytes) for r in range(world)]
            st0 = steps[0][0]
            ring = st0.ring
            assert ring >= 2 and ring % 2 == 0
            for st, _ in steps:
                assert (st.ring, st.buffer, st.parity, st.wait_gathered, st.record_gathered, st.rendered_event) == \
                       (st0.ring, st0.buffer, st0.parity, st0.wait_gathered, st0.record_gathered, st0.rendered_event)
            assert st0.buffer == frame % ring and st0.parity == (frame // (ring // 2)) & 1
            if root < 0:
                for st, recvs in steps:
                    assert st.op == OP_ALLGATHER and not st.in_place and st.render_offset == 0 and not recvs
                continue
            sends = [(r, st.peer) for r, (st, _) in enumerate(steps) if st.op == OP_SEND]
            assert sorted(s for s, _ in sends) == [r for r in range(world) if r != root]
            assert all(dst == root for _, dst in sends)
            st_root, recvs = steps[root]
            if world == 1:
                assert st_root.op == OP_NONE and st_root.in_place and st_root.render_offset == 0
                continue
            assert st_root.op == OP_RECV and st_root.in_place
            assert sorted(p for p, _ in recvs) == [r for r in range(world) if r != root]            # one receive per sender
            slots = sorted([off for _, off in recvs] + [st_root.render_offset])
            assert slots == [k * nbytes for k in range(world)]                                       # disjoint, tile the buffer
            assert all(off == p * nbytes for p, off in recvs) and st_root.render_offset == root * nbytes
            for r, (st, rv) in enumerate(steps):
                if r != root:
                    assert not st.in_place and st.render_offset == 0 and not rv


@pytest.mark.parametrize("world,root", [(2, 0), (8, 0), (8, -1), (4, 3)])
def test_ring_buffers_are_not_rewritten_before_their_gather_drained(world, root):
    """two in-order streams per rank; events as hipEventRecord / hipStreamWaitEvent: a wait observes the latest record that
    was ENQUEUED before it.  Completion times: an operation completes after everything before it on its stream and after the
    events it waited for."""
    nbytes = 4096
    for rank in range(world):
        ring = plan(0, rank, world, root, nbytes)[0].ring
        t_compute = 0.0                     # completion time of the last operation on each stream
        t_comm = 0.0
        rec_time = {}                       # event -> completion time of its latest enqueued record
        exchange_done = {}                  # frame -> completion time of its exchange on the communication stream
        for frame in range(4 * ring + 3):
            st, _ = plan(frame, rank, world, root, nbytes)
            start = t_compute
            if st.wait_gathered:
                assert ("gathered", st.parity) in rec_time, "waits for an event that was never recorded"
                start = max(start, rec_time[("gathered", st.parity)])
            if frame >= ring:
                # the kernel writes buffer `frame % ring`: the exchange that last read it must be over by then
                assert exchange_done[frame - ring] <= start, "frame %d overwrites a buffer whose gather (frame %d) is still running" % (frame, frame - ring)
            t_compute = start + 1.0                                        # the kernel
            rec_time[("rendered", st.rendered_event)] = t_compute
            t_comm = max(t_comm, rec_time[("rendered", st.rendered_event)]) + 7.5   # an exchange much slower than a kernel
            exchange_done[frame] = t_comm
            if st.record_gathered:
                rec_time[("gathered", st.parity)] = t_comm


def test_plan_rejects_bad_arguments():
    st = Step()
    L = N.lib()
    L.vpt_gather_plan.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_uint64, C.POINTER(Step)]
    assert L.vpt_gather_plan(0, 2, 2, 0, 16, C.byref(st)) != 0       # rank outside the world
    assert L.vpt_gather_plan(0, 0, 2, 2, 16, C.byref(st)) != 0       # root outside the world
    assert L.vpt_gather_plan(0, 0, 0, -1, 16, C.byref(st)) != 0
