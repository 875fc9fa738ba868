"""not-gpu: the C-ABI library builds, loads and exports every symbol include/ripor_hip.h declares;
host-only entry points agree with the oracle; without a GPU the product path fails loudly."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    ge.build()
    from ripor_amd import _lib
    return _lib.load()


def test_exports_every_declared_symbol(lib):
    from ripor_amd import _lib
    hdr = open(os.path.join(REPO, "include", "ripor_hip.h")).read()
    declared = set(re.findall(r"\b(rpr_[a-z_0-9]+)\s*\(", hdr))
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in ripor_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert lib.rpr_abi_version() == _lib.ABI_VERSION


def test_struct_layout_matches_header():
    from ripor_amd import _lib
    assert C.sizeof(_lib.KernelStats) == 32
    assert C.sizeof(_lib.DebugTaps) == 6 * C.sizeof(C.c_void_p)
    # 12 int32 + float (52 bytes, padded to 56) + 9 + 15 pointers
    assert C.sizeof(_lib.ModelDesc) == 56 + 24 * 8


def test_rel_bucket_table_matches_torch_expression(lib):
    from oracle import t5_ref
    enc = t5_ref.bucket_table(True, 256)
    dec = t5_ref.bucket_table(False, 64)
    for rel in range(-255, 256):
        assert lib.rpr_rel_bucket(rel, 1, 32, 128) == int(enc[rel + 255]), rel
    for n in range(64):
        assert lib.rpr_rel_bucket(-n, 0, 32, 128) == int(dec[n]), n
    assert lib.rpr_rel_bucket(5, 0, 32, 128) == 0  # future positions collapse to bucket 0 in the decoder


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_gpu_fails_loudly(lib):
    from ripor_amd import engine as E
    from ripor_amd._lib import RiporHipError
    h = C.c_void_p()
    rc = lib.rpr_init(0, C.byref(h))
    assert rc == -3 and b"no CPU fallback" in lib.rpr_last_error()
    with pytest.raises(RiporHipError):
        E.Context.get(0)
    # the Python mirror refuses to search on the CPU instead of silently computing elsewhere
    from ripor_amd.modeling.t5_generative_retriever import T5SeqAQEncoder
    from ripor_amd.tasks.generation import PrefixConstrainLogitProcessorFastSparse, generate_for_constrained_prefix_beam_search
    from ripor_amd.utils import synth
    dims = synth.mini_dims(L=4, enc_layers=1, d_ff=64, vocab_size=64)
    model = T5SeqAQEncoder.from_synthetic(dims)
    proc = PrefixConstrainLogitProcessorFastSparse.from_codes(synth.make_codes(10, 4, 256), 256)
    with pytest.raises(RiporHipError):
        generate_for_constrained_prefix_beam_search(model.base_model, proc, input_ids=torch.ones((1, 4), dtype=torch.long),
                                                    attention_mask=torch.ones((1, 4), dtype=torch.long),
                                                    max_new_tokens=4, num_beams=2, num_return_sequences=2,
                                                    return_dict_in_generate=True, output_scores=True)


def test_environment_variables_the_product_library_reads():
    """The product library reads eight environment variables (README): the documented knobs and the three selectors the
    GPU suite compares bit for bit with the default path. Every other switch of the sources (A/B routes, kernel
    generations, tuning constants, traces) goes through dev_getenv (csrc/common.h), which is compiled out of the product
    build — its name is not even in the binary. The development build of the same sources carries them all."""
    import __graft_entry__ as ge

    def names(path):
        return set(re.findall(rb"RPR_[A-Z][A-Z0-9_]+", open(path, "rb").read()))

    got = {n.decode() for n in names(ge.LIB)}
    assert got == {"RPR_PRECISION", "RPR_FORCED_TAIL", "RPR_FORK_DEPTHS", "RPR_LANE_MIN_ROWS", "RPR_TRIE_THREADS", "RPR_SELECT_RADIX",
                   "RPR_SELECT_LEVELS", "RPR_TAIL_RANK_REPLAY"}, sorted(got)
    dev = {n.decode() for n in names(ge.LIB_DEV)}
    assert got < dev and len(dev) >= 40, sorted(dev)
    srcs = b"".join(open(os.path.join(ge.CSRC, f), "rb").read() for f in ge.SOURCES)
    in_src = {m.decode() for m in re.findall(rb'getenv\("(RPR_[A-Z0-9_]+)"\)', srcs)}
    assert in_src <= dev, sorted(in_src - dev)


def test_product_path_never_imports_oracle():
    """The oracle is test infrastructure: nothing under ripor_amd/ or t5_pretrainer/ may import it."""
    bad = []
    for root in ("ripor_amd", "t5_pretrainer"):
        for dp, _, files in os.walk(os.path.join(REPO, root)):
            for f in files:
                if f.endswith(".py"):
                    src = open(os.path.join(dp, f)).read()
                    if re.search(r"^\s*(from|import)\s+oracle\b", src, re.M):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_docid_to_smtid_streaming_reader(lib, tmp_path):
    """rpr_d2s_* (host only): same docids / codes as json.load on the reference's file format
    (create_customized_smtid_file.py:47-59), whitespace variants included; malformed files are refused."""
    import json
    from ripor_amd import engine as E
    from ripor_amd._lib import RiporHipError
    from ripor_amd.utils import synth
    codes = synth.make_codes(500, 8, 1024, seed=5)
    d2s = {str(1000 + 3 * i): [-1] + [int(x) for x in row] for i, row in enumerate(codes)}
    for k, text in enumerate((json.dumps(d2s), json.dumps(d2s, indent=2), json.dumps(d2s, separators=(",", ":")))):
        p = tmp_path / f"d2s_{k}.json"
        p.write_text(text)
        docids, got = E.read_docid_to_smtid(str(p))
        assert docids == list(d2s.keys())
        assert got.dtype == np.uint16 and np.array_equal(got, codes.astype(np.uint16))
    bad = {
        "ragged": '{"1": [-1, 2, 3], "2": [-1, 4]}',
        "no_minus_one": '{"1": [0, 2, 3]}',
        "negative_code": '{"1": [-1, -2, 3]}',
        "too_large": '{"1": [-1, 70000]}',
        "truncated": '{"1": [-1, 2, 3], "2": [-1, 4',
        "escaped_key": '{"a\\\\"b": [-1, 2]}',
        "empty": "{}",
    }
    for name, text in bad.items():
        p = tmp_path / f"bad_{name}.json"
        p.write_text(text)
        with pytest.raises(RiporHipError):
            E.read_docid_to_smtid(str(p))
    with pytest.raises(RiporHipError):
        E.read_docid_to_smtid(str(tmp_path / "missing.json"))


def test_hot_gemm_kernels_use_no_scratch():
    """The 256 x 256 ping-pong GEMM and the skinny GEMM must compile without scratch: twice in round 4 (and once in round 2)
    a harmless-looking edit made hipcc keep the accumulators or a private copy of the 336-byte argument struct in scratch —
    correct results, +20 % per launch, no warning. Cross-compiles gemm_h2.hip for gfx950 (no GPU needed, ~1 min)."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(REPO, "ripor_amd", "csrc", "gemm_h2.hip")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                        "-Rpass-analysis=kernel-resource-usage", "-o", os.devnull, src], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    seen = {}
    name = None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and name:
            seen[name] = int(m.group(1))
    # every ping-pong instantiation (search and grouped), the skinny kernels (32- and 16-row) and the FULL-tile instantiations
    # of the 128-row LDS-DMA kernel (two definitions of this test once shadowed each other: the second one had dropped these)
    hot = {k: v for k, v in seen.items() if "gemm_h2_pp_" in k or "gemm_h2_skinny" in k or ("gemm_h2_dma_kernel" in k and "Lb1E" in k)}
    assert len(hot) >= 8 and any("skinny16" in k for k in hot) and any("gemm_h2_dma_kernel" in k for k in hot), sorted(seen)
    assert all(v == 0 for v in hot.values()), {k: v for k, v in hot.items() if v}
