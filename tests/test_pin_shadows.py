"""CPU shadows of GPU-only test bodies (tests/test_parity_pins.py): the same functions with the twin library standing in
for the device, at sizes the CPU finishes in seconds.  They do not pin the kernels — twin against twin is an identity as far
as the step goes — they keep the TEST BODIES honest: a host-side change (Archive, loader, spawner, getters) that makes one of
them stale fails here, in `-m "not gpu"`, instead of on the round's last GPU run (round 3 ended red exactly that way:
snapshot() began to restart the route cursor like the reference's Router copy constructor, router.cpp:11-14, and the
one-sided `tw.load(hip.snapshot())` of the 60x60 pin went stale unseen)."""
import test_parity_pins as pins


def test_large_checkpoint_body_on_the_twin(mod, scen, workdir):
    pins.large_checkpoint_body(mod, scen, workdir, lambda c: pins._twin_device(mod, c), 6, 1200, 1500, 8, "auto", True,
                               build_steps=200)


def test_large_checkpoint_body_on_the_twin_fixed_lights(mod, scen, workdir):
    pins.large_checkpoint_body(mod, scen, workdir, lambda c: pins._twin_device(mod, c), 5, 1200, 1000, 12, "ring", False,
                               build_steps=150)


def test_ring_growth_body_on_the_twin(mod, scen, workdir):
    pins.ring_growth_body(mod, scen, workdir, pins._twin_device, steps=120)


def test_many_spawns_body_on_the_twin(mod, scen, workdir):
    pins.many_spawns_body(mod, scen, workdir, pins._twin_device, layouts=("ring",))
