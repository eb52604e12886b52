"""CPU checks of bench.py's reporting helpers (the timed part needs the GPU): the roofline object
carries every field the contract names for both arithmetics, and the profiled-traffic lookup picks
the newest committed summary of the right precision."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_roofline_fields_and_arithmetic():
    import bench
    rays = 640000
    for precision, peak, per in (("fp32", 157.3, 1), ("f16x3", 2516.6, 3)):
        r = bench.roofline(precision, rays, kernel_ms=100.0, launches=5, with_traffic=True)
        for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert key in r
        flop = rays * 127 * bench.FLOP_PER_SAMPLE
        assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == peak
        assert abs(r["achieved"] - flop / 0.1 / 1e12) < 1e-9
        assert abs(r["frac"] - r["achieved"] / peak) < 1e-12
        assert abs(r["executed_frac"] - per * r["frac"]) < 1e-12
        assert r["algorithmic_bytes"] == rays * 212 + 48
        assert r["traffic"] is not None and r["traffic"] < 2 * r["algorithmic_bytes"]
        assert ("f16x3" in r["traffic_source"]) == (precision == "f16x3")
        json.dumps(r)


def test_bench_constants_match_the_survey():
    import bench
    assert bench.FLOP_PER_SAMPLE == 2 * (96 * 256 + 4 * 256 * 256 + 256 * 54)
    assert (bench.IMAGE, bench.SAMPLES) == (800, 128)


def test_defaults_are_the_reference_arithmetic_and_the_one_frame_partition(monkeypatch):
    """The driver-checked line: fp32 (the reference's arithmetic) and, for N > 1, row blocks of ONE
    frame (BASELINE config 4), 100 rows per GPU at N = 8."""
    import bench
    src = open(bench.__file__).read()
    assert 'os.environ.get("NERF_BENCH_PRECISION", "fp32")' in src
    assert 'choices=("weak", "strong"), default="strong"' in src
    assert [bench.shard_rows(r, 8) for r in (0, 3, 7)] == [(0, 100), (300, 400), (700, 800)]
    for world in (1, 2, 3, 4, 8):
        blocks = [bench.shard_rows(r, world) for r in range(world)]
        assert blocks[0][0] == 0 and blocks[-1][1] == bench.IMAGE
        assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
    # one partition for the product and the bench: same blocks, and the per-rank ray counts the
    # line reports (`rays_per_gpu`) add up to the frame for every N the driver runs
    from nerf_amd import parallel
    for world in (1, 2, 4, 8):
        blocks = [bench.shard_rows(r, world) for r in range(world)]
        assert blocks == [parallel.shard_rows(bench.IMAGE, r, world) for r in range(world)]
        rays = [(e - b) * bench.IMAGE for b, e in blocks]
        assert sum(rays) == bench.IMAGE * bench.IMAGE and max(rays) - min(rays) <= bench.IMAGE
        assert rays[0] == bench.IMAGE * bench.IMAGE // world
    assert "--allow-gloo" in src and "refusing to measure" in src      # N > 1 without RCCL fails loudly
    assert bench.PRECISIONS["fp32"]["dtype"] == "f32"
    assert bench.physical_cores() >= 1
    assert isinstance(bench.cpu_model(), str) and bench.cpu_model()
