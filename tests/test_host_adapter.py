"""The C++ adapter systems (engine signature void(World&, float, void*)): they build on CPU, and on a
GPU they leave the engine's state exactly as the oracle says the original systems would."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "sc_gameengine_amd", "host")
BIN = os.path.join(ROOT, "tests", "host", "test_adapter")


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"])
    subprocess.check_call(["make", "-s", "-C", HOST])


def test_adapter_builds_against_the_api_mirror():
    build()
    assert os.path.exists(os.path.join(HOST, "libsc_tick_systems.a")) and os.path.exists(BIN)
    out = subprocess.check_output(["nm", "-C", os.path.join(HOST, "libsc_tick_systems.a")]).decode()
    for sym in ("sc_amd::TransformSystem(sc::World&, float, void*)", "sc_amd::CullingSystem(sc::World&, float, void*)",
                "sc_amd::RenderPrepStreamingSystem(sc::World&, float, void*)", "sc_amd::CreateTickAdapter(int, unsigned int)"):
        assert sym in out, sym


def test_adapter_builds_against_the_real_engine_headers_when_present():
    if not os.path.isdir("/root/reference/src/core/include"):
        pytest.skip("reference tree not on this machine")
    out = subprocess.check_output(["make", "-C", os.path.join(ROOT, "oracle"), "adapter-check"]).decode()
    assert "adapter compiles against the real reference headers" in out


@pytest.mark.gpu
def test_adapter_systems_match_the_oracle_frame_by_frame():
    if not os.path.exists(BIN):
        build()
    r = subprocess.run([BIN, "20000"], capture_output=True, text=True, timeout=300)
    print(r.stdout[-3000:], r.stderr[-2000:])
    assert r.returncode == 0 and "all checks passed" in r.stdout


TILE_BIN = os.path.join(ROOT, "tests", "host", "test_tile_host")


def test_native_tile_host_builds():
    build()
    assert os.path.exists(TILE_BIN)


@pytest.mark.gpu
def test_native_tile_host_runs_the_library_owned_exchange():
    """A C++-only host: communicator from the library (loop-back on one GPU), pipelined scTickTileStep, checked against the oracle."""
    if not os.path.exists(TILE_BIN):
        build()
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([TILE_BIN, "32", "40"], capture_output=True, text=True, timeout=300, env=env)
    print(r.stdout[-3000:], r.stderr[-2000:])
    assert r.returncode == 0 and "all checks passed" in r.stdout
