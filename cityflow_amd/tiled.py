"""One road network over several GPUs: one process per GPU, one tile of intersections per process.

`DistributedEngine` is the `torch.distributed` face of the C++ tiling host (csrc/host/tile_engine.h): every rank
loads the same config, runs the same host spawner (same mt19937 stream, so vehicle ids, priorities and waiting
queues agree without communication), steps its own tile on its own GPU and exchanges the one-lane ghost halo with
its neighbour tiles once per step.  Results are bit-identical to the same network on a single engine
(tests/test_tiling.py).

The halo of a tile is a few KiB per neighbour.  Two transports:
  * mailboxes (default when all ranks share one node): every directed neighbour message has a mailbox in POSIX shared
    memory that both processes map and register with their GPU; the export kernel writes the message straight into
    it and publishes the step's epoch, the import kernel of the neighbour waits for that epoch (`cfx_halo_post` /
    `cfx_halo_wait`).  The step never synchronises with the host: no collective, no copy engine, ~10 us per exchange;
  * staged: `cfx_halo_export` / `cfx_halo_import` through host buffers and a batch of point-to-point messages on a
    host-side (gloo) process group — works across nodes.
`torch.distributed`'s default backend (RCCL on GPUs) carries the reductions of the getters.  The reference has no counterpart (its parallelism is a thread pool inside one
address space, reference src/engine/engine.cpp:19-31).

    torchrun --nproc-per-node 8 my_rl.py        # inside: eng = DistributedEngine(cfg, rows=2, cols=4)
"""
import datetime
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from . import _cityflow


class DistributedEngine:
    def __init__(self, config_file, rows, cols, backend_library="", halo_group=None, mailboxes=None):
        if not dist.is_initialized():
            raise RuntimeError("DistributedEngine needs an initialised torch.distributed process group")
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        if rows * cols != self.world:
            raise ValueError("rows * cols must equal the world size (one tile per process)")
        self._eng = _cityflow.TiledEngine(config_file, rows, cols, [self.rank], backend_library)
        # host-side group for the halo (the buffers live in host memory); reuse the default group if it is gloo
        if halo_group is not None:
            self._halo = halo_group
        elif dist.get_backend() == "gloo":
            self._halo = dist.group.WORLD
        else:  # bounded waits: a rank that failed to set up must not leave the others in a barrier for the default 30 min
            self._halo = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=300))
        self._peers = self._eng.peers(0)
        self._send = torch.from_numpy(self._eng.send_buffer(0))  # zero-copy views of the C++ staging buffers
        self._recv = torch.from_numpy(self._eng.recv_buffer(0))
        self._device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
        self._eng._set_status_reducer(self._reduce_status)
        self._n_lanes = len(self._eng.lane_ids())
        if mailboxes is None:  # shared memory needs one node
            mailboxes = int(os.environ.get("LOCAL_WORLD_SIZE", self.world)) == self.world
        self.mailboxes = bool(mailboxes)
        if self.mailboxes:
            job = [("%d_%d" % (os.getpid(), int(time.time() * 1e3))) if self.rank == 0 else None]
            dist.broadcast_object_list(job, src=0, group=self._halo)
            self._eng.enable_mailboxes(job[0])
            dist.barrier(group=self._halo)   # every process has mapped its mailboxes ...
            self._eng.unlink_mailboxes()     # ... so the names can go

    # ---- stepping -------------------------------------------------------------------------------------------
    def halo_transport(self):
        """How the per-step halo travels between the tiles (bench.py reports it in config.halo)."""
        return "gpu-written shared-memory mailboxes" if self.mailboxes else "staged over gloo"

    def next_step(self):
        self._eng.step_begin()  # spawn, the step's kernels, halo export
        if self.mailboxes:      # device-initiated exchange: nothing for the host to do
            self._eng.step_end()
            return
        ops = []
        for peer, so, sb, ro, rb in self._peers:
            ops.append(dist.P2POp(dist.isend, self._send[so:so + sb], peer, group=self._halo))
            ops.append(dist.P2POp(dist.irecv, self._recv[ro:ro + rb], peer, group=self._halo))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        self._eng.step_end()  # halo import (asynchronous on the tile's stream)

    # ---- reference API subset (every rank gets the whole-network answer) ----------------------------------
    def _sum(self, arr, dtype):
        t = torch.as_tensor(np.ascontiguousarray(arr), dtype=dtype).to(self._device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.cpu().numpy()

    def get_lane_vehicle_count_array(self):
        return self._sum(self._eng.get_lane_vehicle_count_array(), torch.int32)

    def get_lane_waiting_vehicle_count_array(self):
        return self._sum(self._eng.get_lane_waiting_vehicle_count_array(), torch.int32)

    def lane_ids(self):
        return self._eng.lane_ids()

    def get_lane_vehicle_count(self):
        return dict(zip(self._eng.lane_ids(), self.get_lane_vehicle_count_array().tolist()))

    def get_lane_waiting_vehicle_count(self):
        return dict(zip(self._eng.lane_ids(), self.get_lane_waiting_vehicle_count_array().tolist()))

    def scalars(self):
        s = self._eng._scalars()
        ints = self._sum([s["active_vehicle_count"], s["finished_vehicle_count"], s["vehicle_steps"]], torch.int64)
        tt = self._sum([s["cumulative_travel_time"]], torch.float64)
        return {"step": s["step"], "spawned_vehicle_count": s["spawned_vehicle_count"],
                "active_vehicle_count": int(ints[0]), "finished_vehicle_count": int(ints[1]),
                "vehicle_steps": int(ints[2]), "cumulative_travel_time": float(tt[0])}

    def local_scalars(self):
        return self._eng._scalars()

    def get_vehicle_count(self):
        return self.scalars()["active_vehicle_count"]

    def get_current_time(self):
        return self._eng.get_current_time()

    def set_tl_phase(self, intersection_id, phase_id):
        self._eng.set_tl_phase(intersection_id, phase_id)  # every rank makes the same calls; the owner applies them

    def set_tl_phases(self, phases):
        self._eng.set_tl_phases(phases)

    def reset(self, seed=False):
        self._eng.sync()
        dist.barrier(group=self._halo)  # nobody reuses a mailbox buffer a neighbour has not consumed yet
        self._eng.reset(seed)
        dist.barrier(group=self._halo)

    def sync(self):
        self._eng.sync()

    def local_vehicle_state(self):
        return self._eng._vehicle_state()

    # ---- string-keyed getters: every rank contributes the vehicles of its tile (an all-gather of small dicts; meant
    #      for inspection, the array getters above are the RL path)
    def _merged(self, local):
        parts = [None] * self.world
        dist.all_gather_object(parts, local, group=self._halo)
        out = {}
        for p in parts:
            out.update(p)
        return out

    def get_vehicle_speed(self):
        return self._merged(self._eng.get_vehicle_speed())

    def get_vehicle_distance(self):
        return self._merged(self._eng.get_vehicle_distance())

    def get_lane_vehicles(self):
        parts = [None] * self.world
        dist.all_gather_object(parts, {k: v for k, v in self._eng.get_lane_vehicles().items() if v}, group=self._halo)
        out = {k: [] for k in self._eng.lane_ids()}
        for p in parts:
            out.update(p)  # a lane belongs to exactly one tile
        return out

    def push_vehicle(self, info, roads):
        self._eng.push_vehicle(info, roads)  # every rank makes the same call (the spawners stay identical)

    def set_vehicle_speed(self, vehicle_id, speed):
        self._eng.set_vehicle_speed(vehicle_id, speed)

    # a priority collision in the spawner asks "has vehicle X finished?"; every rank asks at the same point of the
    # same RNG stream, so a collective is legal here and keeps the streams identical
    def _reduce_status(self, status):
        t = torch.tensor([int(status)], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self._halo)
        return int(t[0])
