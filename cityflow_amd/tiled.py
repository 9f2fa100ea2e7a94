"""One road network over several GPUs: one process per GPU, one tile of intersections per process.

`DistributedEngine` is the `torch.distributed` face of the C++ tiling host (csrc/host/tile_engine.h): every rank
loads the same config, runs the same host spawner (same mt19937 stream, so vehicle ids, priorities and waiting
queues agree without communication), steps its own tile on its own GPU and exchanges the one-lane ghost halo with
its neighbour tiles once per step.  Results are bit-identical to the same network on a single engine
(tests/test_tiling.py).

The halo of a tile is a few KiB per neighbour.  Transports (`transport=`; None tries them in this order and takes the
first one EVERY rank can set up):
  * "device": mailboxes in the RECEIVING tile's device memory — peer HBM, written over xGMI by the sender's export
    kernel (hipIpc between the processes of a node); the export publishes the step's epoch, the neighbour's import kernel
    waits for it (`cfx_halo_post` / `cfx_halo_wait`).  The step never synchronises with the host: no collective, no copy
    engine;
  * "host": the same protocol with the mailboxes in POSIX shared memory both processes map and register with their GPU
    (PCIe instead of xGMI);
  * "rccl": `cfx_halo_export` leaves the messages in device memory, one batch of RCCL send / recv (torch.distributed
    P2P on the default nccl group) moves them GPU to GPU, `cfx_halo_import` reads them there — one host round trip per
    step, works across nodes;
  * "gloo": the messages staged through host buffers and a host-side gloo group — works everywhere.
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


class _DeviceMemory:
    """Raw device memory owned by the engine, for torch.as_tensor (CUDA array interface)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def _device_bytes(ptr, nbytes, device):
    if nbytes == 0:
        return torch.empty(0, dtype=torch.uint8, device=device)
    if device.type == "cpu":
        import ctypes
        return torch.from_numpy(np.ctypeslib.as_array((ctypes.c_ubyte * nbytes).from_address(ptr)))
    return torch.as_tensor(_DeviceMemory(ptr, nbytes), device=device)


class DistributedEngine:
    def __init__(self, config_file, rows, cols, backend_library="", halo_group=None, mailboxes=None, transport=None):
        if not dist.is_initialized():
            raise RuntimeError("DistributedEngine needs an initialised torch.distributed process group")
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        if rows * cols != self.world:
            raise ValueError("rows * cols must equal the world size (one tile per process)")
        built_error = None
        try:
            self._eng = _cityflow.TiledEngine(config_file, rows, cols, [self.rank], backend_library)
        except Exception as exc:  # noqa: BLE001 - agreed on below: every rank must leave the constructor the same way
            built_error = str(exc)[:300]
        # host-side group for the halo (the buffers live in host memory); reuse the default group if it is gloo
        if halo_group is not None:
            self._halo = halo_group
        elif dist.get_backend() == "gloo":
            self._halo = dist.group.WORLD
        else:  # bounded waits: a rank that failed to set up must not leave the others in a barrier for the default 30 min
            self._halo = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=300))
        if not self._all_ok(built_error is None):  # (a rank that raised alone would leave the others in the collectives below)
            raise RuntimeError("DistributedEngine: the tile could not be built on every rank" +
                               (": " + built_error if built_error else ""))
        self._peers = self._eng.peers(0)
        self._send = torch.from_numpy(self._eng.send_buffer(0))  # zero-copy views of the C++ staging buffers
        self._recv = torch.from_numpy(self._eng.recv_buffer(0))
        self._device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
        self._eng._set_status_reducer(self._reduce_status)
        self._n_lanes = len(self._eng.lane_ids())
        one_node = int(os.environ.get("LOCAL_WORLD_SIZE", self.world)) == self.world
        staged = "rccl" if (dist.get_backend() == "nccl" and backend_library == "") else "gloo"
        if transport is not None:
            candidates = [transport]
        elif mailboxes is not None:  # (the older switch: mailboxes or staged)
            candidates = ["device", "host"] if mailboxes else [staged]
        else:
            candidates = (["device", "host"] if one_node else []) + ([staged, "gloo"] if staged != "gloo" else ["gloo"])
        self.transport = None
        errors = []
        for cand in candidates:
            try:
                ok = self._setup(cand)
            except Exception as exc:  # noqa: BLE001 - any local failure disqualifies the transport for everybody
                ok = False
                errors.append("%s: %s" % (cand, str(exc)[:160]))
            if self._all_ok(ok):
                self.transport = cand
                break
            stuck = torch.tensor([1 if self._eng.halo_transport() != "staged" else 0], dtype=torch.int32)
            dist.all_reduce(stuck, op=dist.ReduceOp.MAX, group=self._halo)
            if int(stuck.item()):
                # some rank attached its mailboxes and another could not: these engines cannot go back (every rank raises)
                raise RuntimeError("halo transport %r could not be set up on every rank (%s); build a new DistributedEngine "
                                   "with another transport" % (cand, "; ".join(errors)))
        if self.transport is None:
            raise RuntimeError("no halo transport could be set up: " + "; ".join(errors))
        self.mailboxes = self.transport in ("device", "host")

    def _all_ok(self, ok):
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self._halo)
        return int(flag.item()) == 1

    def _all_on_one_device(self):
        """True if every tile of every rank reports the same PHYSICAL device (cfx_device_identity: the PCI bus id; the CPU
        twin says "cpu").  A collective: every rank calls it at the same point."""
        mine = list(self._eng.device_identities())
        everyone = [None] * self.world
        dist.all_gather_object(everyone, mine, group=self._halo)
        ids = {i for part in everyone for i in part}
        return len(ids) == 1 and "?" not in ids

    def _job_name(self):
        job = [("%d_%d" % (os.getpid(), int(time.time() * 1e3))) if self.rank == 0 else None]
        dist.broadcast_object_list(job, src=0, group=self._halo)
        return job[0]

    def _setup(self, transport):
        # Every rank runs the same sequence of collectives whatever happens locally: a local failure becomes ok = False
        # and is agreed on afterwards (a rank that raised between two collectives would leave the others waiting).
        def local(fn, *a):
            try:
                r = fn(*a)
                return True if r is None else bool(r)
            except Exception as exc:  # noqa: BLE001
                self._setup_error = str(exc)[:160]
                return False

        if transport == "device":
            job = self._job_name()
            ok = local(self._eng.device_mailbox_phase, job, 1)
            one_device = self._all_on_one_device()  # (a collective: every rank, whatever happened locally)
            if ok and not self._eng.device_mailboxes_fine_grained():
                # plain device memory behind a mailbox (the platform would not export fine-grained memory): a kernel on
                # ANOTHER GPU is not guaranteed to see the sender's stores while it runs — fine only on one shared device
                # (decided from the engines, not from torch: the process group may be gloo over HIP engines, and with a
                # per-rank HIP_VISIBLE_DEVICES every rank sees "one device" while sitting on a different GPU)
                ok = one_device
            if not self._all_ok(ok):  # nobody has attached anything yet: the next transport can still be tried
                local(self._eng.unlink_mailboxes)
                return False
            ok = local(self._eng.device_mailbox_phase, job, 2)
            dist.barrier(group=self._halo)
            local(self._eng.unlink_mailboxes)
            return ok
        if transport == "host":
            job = self._job_name()
            ok = local(self._eng.enable_mailboxes, job)
            dist.barrier(group=self._halo)   # every process has mapped its mailboxes ...
            local(self._eng.unlink_mailboxes)  # ... so the names can go
            return ok
        if transport == "rccl":
            if dist.get_backend() != "nccl" and self._device.type != "cpu":
                raise RuntimeError("the rccl transport needs the nccl (RCCL) process group")
            # (a CPU engine's "device" buffers are host memory and the default group is gloo: same code path, for tests)
            if self._device.type != "cpu":
                # the tensors below are labelled with torch's current device: it must be the one the engine's buffers are on
                want = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
                if torch.cuda.current_device() != want:
                    raise RuntimeError("the rccl transport needs torch.cuda.set_device(LOCAL_RANK %% device_count) before the "
                                       "engine is built (current device %d, engine on %d)" % (torch.cuda.current_device(), want))
            sp, sn, rp, rn = self._eng.halo_device_buffers(0)
            self._dsend = _device_bytes(sp, sn, self._device)
            self._drecv = _device_bytes(rp, rn, self._device)
            return True
        if transport == "gloo":
            return True
        raise ValueError("unknown halo transport %r" % (transport,))

    # ---- stepping -------------------------------------------------------------------------------------------
    def halo_transport(self):
        """How the per-step halo travels between the tiles (bench.py reports it in config.halo)."""
        return {"device": "gpu-written mailboxes in the receiving GPU's HBM (hipIpc peer memory, xGMI)",
                "host": "gpu-written mailboxes in shared host memory",
                "rccl": "RCCL send/recv of device-resident messages, one batch per step",
                "gloo": "staged through host buffers over gloo"}[self.transport]

    def _replay(self):
        # saveReplay with one tile per process: every rank's part of the step's line goes to rank 0, which writes it
        if not self._eng._wants_replay():
            return
        parts = [None] * self.world if self.rank == 0 else None
        dist.gather_object(self._eng._replay_part(), parts, dst=0, group=self._halo)
        if self.rank == 0:
            self._eng._replay_write(parts)

    def set_save_replay(self, open):
        self._eng.set_save_replay(open)

    def set_replay_file(self, replay_file):
        self._eng.set_replay_file(replay_file)

    def next_step(self):
        self._step()
        if self.world > 1:  # (one process running every tile writes its replay line itself, TiledEngineHost::stepEnd)
            self._replay()
            # the finished vehicles are forgotten once enough numbers are out (EngineHost::compactVehicles over tiles): every
            # rank runs the whole spawner, so every rank says so after the same step
            if self._eng._wants_compaction():
                self.compact_vehicles()

    def compact_vehicles(self):
        """Forget the finished vehicles now (the reference frees a vehicle when it finishes, engine.cpp:296-310): a collective —
        every rank's part of the state goes to every rank, which renumbers the vehicles alive and keeps its tile's part."""
        self._eng.sync()
        parts = [None] * self.world
        dist.all_gather_object(parts, self._eng._snapshot_part(), group=self._halo)
        dist.barrier(group=self._halo)  # nobody reuses a mailbox buffer a neighbour has not consumed yet (as load())
        self._eng._compact_from_parts(parts)
        dist.barrier(group=self._halo)

    def _step(self):
        if self.transport == "rccl":
            self._eng.step_begin_device()  # spawn, the step's kernels, halo export; returns when the messages are complete
            ops = []
            for peer, so, sb, ro, rb in self._peers:
                ops.append(dist.P2POp(dist.isend, self._dsend[so:so + sb], peer))
                ops.append(dist.P2POp(dist.irecv, self._drecv[ro:ro + rb], peer))
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
                if self._device.type != "cpu":
                    torch.cuda.current_stream().synchronize()  # the import runs on the engine's own stream
            self._eng.step_end_device()
            return
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
        ints = self._sum([s["active_vehicle_count"], s["finished_vehicle_count"], s["vehicle_steps"], s["tie_events"]], torch.int64)
        tt = self._sum([s["cumulative_travel_time"]], torch.float64)
        return {"step": s["step"], "spawned_vehicle_count": s["spawned_vehicle_count"],
                "active_vehicle_count": int(ints[0]), "finished_vehicle_count": int(ints[1]),
                "vehicle_steps": int(ints[2]), "tie_events": int(ints[3]), "cumulative_travel_time": float(tt[0])}

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

    def get_vehicles(self, include_waiting=False):
        parts = [None] * self.world
        dist.all_gather_object(parts, self._eng._vehicles_keyed(include_waiting), group=self._halo)
        keyed = [p for part in parts for p in part]
        if include_waiting:  # pushed since the last step: every rank's spawner holds the same ones, listed once
            keyed += self._eng._pending_pushed_keyed()
        return [vid for _, vid in sorted(keyed)]  # vehiclePool order = by priority (unique)

    def get_vehicle_info(self, vehicle_id):
        """Every rank makes the same call; the rank that runs the vehicle has the details."""
        parts = [None] * self.world
        dist.all_gather_object(parts, self._eng.get_vehicle_info(vehicle_id), group=self._halo)
        return max(parts, key=len)

    def get_leader(self, vehicle_id):
        mine = self._eng.get_leader(vehicle_id)  # raises on every rank alike if the vehicle is unknown or has finished
        parts = [None] * self.world
        dist.all_gather_object(parts, mine if self._eng._runs_here(vehicle_id) else None, group=self._halo)
        found = [p for p in parts if p is not None]
        return found[0] if found else ""

    def get_average_travel_time(self):
        s = self._eng._scalars()
        tt = self._sum([s["cumulative_travel_time"]], torch.float64)
        fin = self._sum([s["finished_vehicle_count"]], torch.int64)
        st = torch.as_tensor(np.ascontiguousarray(self._eng._local_status()), dtype=torch.uint8).to(self._device)
        if st.numel():
            dist.all_reduce(st, op=dist.ReduceOp.MAX)
        return self._eng._average_travel_time_from(float(tt[0]), int(fin[0]), st.cpu().numpy())

    def set_random_seed(self, seed):
        self._eng.set_random_seed(seed)  # every rank alike: the spawners stay identical

    def push_vehicle(self, info, roads):
        self._eng.push_vehicle(info, roads)  # every rank makes the same call (the spawners stay identical)

    def set_vehicle_speed(self, vehicle_id, speed):
        self._eng.set_vehicle_speed(vehicle_id, speed)

    def set_vehicle_route(self, vehicle_id, route):
        """Engine::setRoute: every rank makes the same call (the route tables stay identical); where the vehicle runs is
        merged through the status reducer."""
        return self._eng.set_vehicle_route(vehicle_id, route)

    # ---- archive (reference src/engine/archive.cpp): a snapshot is assembled on every rank from one part per rank; a load
    #      needs no communication — every rank reads the same archive and keeps its tile's part
    def snapshot(self):
        parts = [None] * self.world
        dist.all_gather_object(parts, self._eng._snapshot_part(), group=self._halo)
        return self._eng._snapshot_from_parts(parts)

    def load(self, archive):
        self._eng.sync()
        dist.barrier(group=self._halo)  # nobody reuses a mailbox buffer a neighbour has not consumed yet
        self._eng.load(archive)
        dist.barrier(group=self._halo)

    def load_from_file(self, path):
        self._eng.sync()
        dist.barrier(group=self._halo)
        self._eng.load_from_file(path)
        dist.barrier(group=self._halo)

    # a priority collision in the spawner asks "has vehicle X finished?"; every rank asks at the same point of the
    # same RNG stream, so a collective is legal here and keeps the streams identical
    def _reduce_status(self, status):
        t = torch.tensor([int(status)], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self._halo)
        return int(t[0])
