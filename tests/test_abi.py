"""The C-ABI library: loads, exports every symbol include/nextplaid_hip.h declares, reads the crate's
on-disk format, and FAILS LOUDLY (no CPU fallback) when no GPU is present.  CPU only: no compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from helpers import ROOT, synth

import next_plaid_amd as npa
from next_plaid_amd import api


def test_header_symbols_are_exported():
    hdr = open(os.path.join(ROOT, "include", "nextplaid_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(np_hip_\w+)\s*\(", hdr)))
    assert len(declared) >= 15
    L = api.lib()
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, f"not exported: {missing}"
    assert sorted(api.EXPORTS) == declared, "api.EXPORTS must list exactly the header's entry points"


def test_struct_sizes_match_header(tmp_path):
    # ctypes mirrors vs the C structs as gcc lays them out from include/nextplaid_hip.h (size AND field offsets)
    import subprocess
    names = ["np_open_opts", "np_search_params", "np_stats", "np_info", "np_index_arrays", "np_synth_spec", "np_write_opts"]
    src = ["#include <stdio.h>", "#include <stddef.h>", '#include "nextplaid_hip.h"', "int main(void) {"]
    for n in names:
        src.append(f'  printf("{n} %zu\\n", sizeof({n}));')
        for f, _ in getattr(api, n)._fields_:
            src.append(f'  printf("{n}.{f} %zu\\n", offsetof({n}, {f}));')
    src += ["  return 0;", "}"]
    c = tmp_path / "sz.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)])
    got = dict(l.split() for l in subprocess.check_output([str(exe)], text=True).splitlines())
    for n in names:
        st = getattr(api, n)
        assert C.sizeof(st) == int(got[n]), n
        for f, _ in st._fields_:
            assert getattr(st, f).offset == int(got[f"{n}.{f}"]), f"{n}.{f}"
    assert C.sizeof(api.np_search_params) == 28 and C.sizeof(api.np_open_opts) == 32


def test_no_oracle_in_product_path():
    # the product package must never import / link the oracle
    for dirpath, _, files in os.walk(os.path.join(ROOT, "next-plaid_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "plaid_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f


def test_load_missing_dir_is_index_load_error():
    with pytest.raises(npa.IndexLoadError):
        npa.MmapIndex.load("/nonexistent/index")


def _write(tmp_path, **kw):
    spec = synth.SynthSpec(num_docs=120, num_centroids=32, dim=64, nbits=4, doc_len_min=0, doc_len_max=12, seed=3)
    a = synth.generate_arrays(spec)
    synth.write_index(str(tmp_path), a, chunk_docs=50, **kw)
    return a


def test_loader_parses_reference_layout_then_needs_gpu(tmp_path, gpu_available):
    _write(tmp_path)
    assert sorted(os.listdir(tmp_path))[:3] == ["0.codes.npy", "0.metadata.json", "0.residuals.npy"]
    if gpu_available:
        ix = npa.MmapIndex.load(str(tmp_path))
        assert ix.num_documents() == 120
    else:
        # files parse (no IndexLoad/Shape error), then the missing device is reported, not papered over
        with pytest.raises(npa.DeviceUnavailableError):
            npa.MmapIndex.load(str(tmp_path))


def test_loader_error_codes(tmp_path):
    a = _write(tmp_path)
    os.remove(tmp_path / "bucket_weights.npy")            # codec.rs:428-431
    with pytest.raises(npa.CodecError):
        npa.MmapIndex.load(str(tmp_path))
    np.save(tmp_path / "bucket_weights.npy", a["bucket_weights"])
    np.save(tmp_path / "1.residuals.npy", a["residuals"][:10, :7])   # wrong packed width
    with pytest.raises(npa.ShapeError):
        npa.MmapIndex.load(str(tmp_path))
    (tmp_path / "centroids.npy").write_bytes(b"not an npy file at all")
    with pytest.raises(npa.IndexLoadError):
        npa.MmapIndex.load(str(tmp_path))


def test_loader_accepts_fastplaid_i64_ivf_lengths(tmp_path, gpu_available):
    a = _write(tmp_path)
    np.save(tmp_path / "ivf_lengths.npy", a["ivf_lengths"].astype("<i8"))   # mmap.rs:1780-1789
    exc = None if gpu_available else npa.DeviceUnavailableError
    if exc:
        with pytest.raises(exc):
            npa.MmapIndex.load(str(tmp_path))
    else:
        npa.MmapIndex.load(str(tmp_path))


def test_search_without_gpu_raises(gpu_available):
    if gpu_available:
        pytest.skip("GPU present")
    spec = synth.SynthSpec(num_docs=20, num_centroids=8, dim=64, doc_len_min=4, doc_len_max=4)
    a = synth.generate_arrays(spec)
    with pytest.raises(npa.DeviceUnavailableError):
        npa.MmapIndex.from_arrays(a["centroids"], a["bucket_weights"], a["ivf"], a["ivf_lengths"], a["doc_lengths"],
                                  a["codes"], a["residuals"], a["nbits"])
    assert npa.device_count() == 0


def test_params_mirror_reference_defaults():
    p = npa.SearchParameters()          # search.rs:58-69
    assert (p.batch_size, p.n_full_scores, p.top_k, p.n_ivf_probe, p.centroid_batch_size,
            p.centroid_score_threshold) == (2000, 4096, 10, 8, 100_000, 0.4)
    c = p._c()
    assert c.has_threshold == 1 and abs(c.centroid_score_threshold - 0.4) < 1e-7
    assert api.lib().np_hip_n_sel(C.byref(c)) == 1024          # max(4096/4, 10)
    assert api.lib().np_hip_n_sel(C.byref(npa.SearchParameters(n_full_scores=8, top_k=10)._c())) == 8


def test_one_hip_runtime_per_process_whatever_the_import_order():
    """PyTorch-ROCm bundles its own libamdhip64; if libnextplaid_hip.so dragged in /opt/rocm's copy first, a later
    `import torch` would add a second runtime and torch.cuda would report "no ROCm-capable device".  api.lib()
    therefore loads torch's copy first when torch is installed.  Checked in a fresh interpreter."""
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "from next_plaid_amd import api\n"
        "api.lib()\n"
        "import torch\n"
        "libs = {l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l}\n"
        "print(len(libs), sorted(libs))\n"
        "assert len(libs) == 1, libs\n" % os.path.join(ROOT, "next-plaid_amd"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr


def test_bad_rccl_override_is_an_error_code_not_a_crash():
    """np_dist.hip resolves librccl with dlopen; a candidate that fails to load (here NEXTPLAID_RCCL_LIB pointing nowhere)
    must leave a message and move on to the next candidate -- dlerror() returns NULL on its second call, and assigning that
    to a std::string crashed the process.  Whatever the outcome (a later candidate loads, or none does), the
    entry returns a status code."""
    import subprocess
    import sys
    code = ("import ctypes as C, sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from next_plaid_amd import api\n"
            "L = api.lib(); buf = C.create_string_buffer(128)\n"
            "rc = L.np_hip_comm_unique_id(buf)\n"
            "print('rc', rc, L.np_hip_last_error().decode() if rc else '')\n") % (ROOT, os.path.join(ROOT, "next-plaid_amd"))
    env = dict(os.environ, NEXTPLAID_RCCL_LIB="/nonexistent/librccl.so.9")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    rc = int(r.stdout.split()[1])
    assert rc in (0, 6), r.stdout          # NP_OK (a later candidate loaded) or NP_ERR_DEVICE_UNAVAILABLE
    if rc == 6:
        assert "librccl is not available" in r.stdout or "ncclGetUniqueId" in r.stdout


def test_search_parameters_serde_shape():
    """search.rs:26-48: four required counts, serde defaults for the two newer fields, null threshold = None."""
    P = npa.SearchParameters
    p = P.from_json('{"batch_size": 2000, "n_full_scores": 8192, "top_k": 20, "n_ivf_probe": 16}')
    assert (p.n_full_scores, p.top_k, p.n_ivf_probe, p.centroid_batch_size, p.centroid_score_threshold) == (8192, 20, 16, 100_000, 0.4)
    p = P.from_json('{"batch_size": 1, "n_full_scores": 4, "top_k": 1, "n_ivf_probe": 1, "centroid_score_threshold": null, '
                    '"centroid_batch_size": 0, "unknown": 5}')
    assert p.centroid_score_threshold is None and p.centroid_batch_size == 0
    with pytest.raises(ValueError, match="missing field `top_k`"):
        P.from_json('{"batch_size": 1, "n_full_scores": 4, "n_ivf_probe": 1}')
    with pytest.raises(ValueError, match="usize"):
        P.from_json('{"batch_size": 1, "n_full_scores": -4, "top_k": 1, "n_ivf_probe": 1}')
    q = P.from_json(P().to_json())
    assert q == P() and "precision" not in P().to_json()      # the extra knob never leaks into the crate's JSON
