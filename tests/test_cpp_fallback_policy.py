"""The C++ host mirror honours the crate's accelerator switches (NEXT_PLAID_FORCE_GPU / NEXT_PLAID_FORCE_CPU,
lib.rs:71-84) and the broken-flag hand-off to a CPU path (cuda.rs:52-182): the tested stand-in for the `hip` feature's
Rust wrapper.  Runs on a GPU-less host (DeviceUnavailable is what triggers the hand-off)
and on a GPU box (the device serves the call, FORCE_CPU still skips it)."""
import os
import subprocess

import pytest

from helpers import ROOT, make_arrays, synth


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = tmp_path_factory.mktemp("fb") / "fallback_policy"
    csrc = os.path.join(ROOT, "next-plaid_amd", "csrc")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-o", str(out), os.path.join(ROOT, "tests", "cpp", "fallback_policy.cpp"),
                           "-L", csrc, "-lnextplaid_hip", f"-Wl,-rpath,{csrc}"])
    return str(out)


@pytest.fixture(scope="module")
def index_dir(tmp_path_factory):
    spec, a = make_arrays(num_docs=200, num_centroids=64, dim=64, nbits=4, doc_len_min=3, doc_len_max=12, seed=5)
    p = str(tmp_path_factory.mktemp("idx"))
    synth.write_index(p, a, chunk_docs=100)
    return p


def run(exe, index_dir, hook=1, **env):
    e = {k: v for k, v in os.environ.items() if not k.startswith("NEXT_PLAID_FORCE")}
    e.update(env)
    return subprocess.run([exe, index_dir, "64", str(hook)], env=e, capture_output=True, text=True, timeout=120)


def test_device_failure_hands_off_to_cpu_and_raises_the_flag(exe, index_dir, gpu_available):
    r = run(exe, index_dir)
    lines = r.stdout.strip().splitlines()
    if gpu_available:
        assert lines[0] == "device" and lines[1] == "again device"
    else:
        assert lines[0] == "cpu 1 broken=1", r.stdout + r.stderr       # DeviceUnavailable -> flag -> CPU hook
        assert lines[1] == "again cpu"                                 # flag raised: the device is not retried
        assert "Falling back to CPU" in r.stderr
    assert lines[2] == "cleared broken=0"
    assert lines[3] == "geom 200 64 64", r.stdout          # accessors valid with or without a device handle
    assert lines[4] == ("reloaded device 200" if gpu_available else "reloaded cpu 200"), r.stdout   # index.rs:1767, load()'s policy
    if not gpu_available:
        assert lines[5] == "decompress error 6"            # device-only method: a clear Error, not a NULL handle in the ABI


def test_force_gpu_never_falls_back(exe, index_dir, gpu_available):
    r = run(exe, index_dir, NEXT_PLAID_FORCE_GPU="1")
    first = r.stdout.strip().splitlines()[0]
    assert first == ("device" if gpu_available else "error 6"), r.stdout          # Error::DeviceUnavailable surfaces
    r = run(exe, index_dir, NEXT_PLAID_FORCE_GPU="true", NEXT_PLAID_FORCE_CPU="1")   # FORCE_GPU wins (lib.rs:80-84)
    assert r.stdout.strip().splitlines()[0] == ("device" if gpu_available else "error 6")


def test_force_cpu_skips_the_device(exe, index_dir):
    r = run(exe, index_dir, NEXT_PLAID_FORCE_CPU="TRUE")
    lines = r.stdout.strip().splitlines()
    assert lines[0] == "cpu 1 broken=0" and lines[1] == "again cpu", r.stdout      # no device call, no flag
    assert "Falling back" not in r.stderr


def test_without_a_cpu_path_nothing_is_papered_over(exe, index_dir, gpu_available):
    r = run(exe, index_dir, hook=0)
    assert r.stdout.strip().splitlines()[0] == ("device" if gpu_available else "error 6")
    r = run(exe, "/nonexistent/index", hook=1)
    # a missing index is IndexLoad on a GPU box; without a device the device check comes second, so it is IndexLoad too
    assert r.stdout.strip().splitlines()[0] == "error 1", r.stdout
