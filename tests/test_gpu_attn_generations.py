"""The second-generation fp32-MFMA attention kernels (tail self / cross attention, search encoder attention;
ripor_amd/csrc/tail_kernels.hip) against the kernels they replace, through the product path: the same searches in
subprocesses that load the development build of the library (RPR_DEV_LIB=1) and differ only in RPR_TAIL_ATTN_GEN /
RPR_ENC_ATTN_MFMA / RPR_STEP_CROSS_MFMA (read once, when the library loads). The tail kernels issue the same MFMAs in the same order and must give the same bits; the encoder and the step
cross-attention move from VALU sums to MFMA sums (fp32 rounding order changes): same ranked smtids, scores within 1e-5.
Reference semantics: T5Attention inside t5_pretrainer/modeling/t5_generative_retriever.py (softmax(QK^T + bias) V in fp32)."""
import os
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _dump(tmp_path, name, beams, L=32, **env):
    out = str(tmp_path / (name + ".npz"))
    # RPR_DEV_LIB=1: the switches below are development switches, live only in libripor_hip_dev.so (same sources, -DRPR_DEV_SWITCHES)
    e = dict(os.environ, PYTHONPATH=REPO + os.pathsep + os.environ.get("PYTHONPATH", ""), RPR_DEV_LIB="1", **{k: str(v) for k, v in env.items()})
    p = subprocess.run([sys.executable, os.path.join(REPO, "tools", "attn_gen_dump.py"), out, str(beams), str(L)], cwd=REPO, env=e,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    d = dict(np.load(out))
    d["_stderr"] = p.stderr
    return d


@pytest.mark.parametrize("beams,L", [(10, 32), (33, 32), (10, 16), (7, 9)])
def test_tail_attention_generations_give_the_same_bits(tmp_path, beams, L):
    # (encoder and step cross-attention on their VALU kernels in both runs: those two change the summation order)
    old = _dump(tmp_path, "gen1", beams, L, RPR_TAIL_ATTN_GEN=1, RPR_ENC_ATTN_MFMA=0, RPR_STEP_CROSS_MFMA=0)
    new = _dump(tmp_path, "gen2", beams, L, RPR_TAIL_ATTN_GEN=2, RPR_ENC_ATTN_MFMA=0, RPR_STEP_CROSS_MFMA=0)
    for k in old:
        if k != "_stderr":
            assert old[k].shape == new[k].shape and old[k].tobytes() == new[k].tobytes(), k
    for k in ("fork3", "fork5_7"):     # explicit forks inside the kernels' range (T <= 8) were taken
        assert new[k + "_forks"][:, 1].sum() > 0, k


def test_encoder_and_step_cross_attention_on_the_mfma_tile(tmp_path):
    old = _dump(tmp_path, "valu", 10, RPR_ENC_ATTN_MFMA=0, RPR_STEP_CROSS_MFMA=0)
    enc = _dump(tmp_path, "enc", 10, RPR_ENC_ATTN_MFMA=1, RPR_STEP_CROSS_MFMA=0)
    both = _dump(tmp_path, "both", 10, RPR_ENC_ATTN_MFMA=1, RPR_STEP_CROSS_MFMA=1)
    m16 = _dump(tmp_path, "m16", 10, RPR_ENC_ATTN_MFMA=0, RPR_STEP_CROSS_MFMA=2)   # the 16 x 16 tile kernel of the steps (default)
    assert any(not np.array_equal(m16[k + "_scores"], old[k + "_scores"]) for k in ("auto", "fork3", "fork5_7")), \
        "the 16 x 16 step cross-attention did not run"
    live = old["mask"] != 0
    for new in (enc, both, m16):
        err = np.abs(new["encoder_out"] - old["encoder_out"])[live].max()
        assert err < 2e-5, err
        for k in ("auto", "fork3", "fork5_7"):
            assert (new[k + "_tokens"] == old[k + "_tokens"]).all(), k
            assert np.abs(new[k + "_scores"] - old[k + "_scores"]).max() < 1e-5, k
    assert not np.array_equal(enc["encoder_out"], old["encoder_out"]), "the MFMA encoder attention did not run"


def test_row_split_of_ragged_256_tile_gemm_launches(tmp_path):
    """launch_gemm_h2 hands the rows behind the last whole round of 256 x 256 tiles to the 128 x 128 tile kernel as a second
    launch (beam 1000 with one query: 318 tiles on 256 CUs). Here every launch of the 256-tile route with two or more row tiles
    is split in the middle (RPR_GEMM_ROWSPLIT=2 on top of RPR_GEMM_TILE=256): the packed encoder (live row count on the device),
    the steps' 370 rows, the tail pass with planes / fused-norm / residual-plane epilogues — against the same searches unsplit."""
    whole = _dump(tmp_path, "whole", 10, RPR_GEMM_TILE=256, RPR_GEMM_ROWSPLIT=0, RPR_GEMM_ROWSPLIT_LOG=1)
    split = _dump(tmp_path, "split", 10, RPR_GEMM_TILE=256, RPR_GEMM_ROWSPLIT=2, RPR_GEMM_ROWSPLIT_LOG=1)
    assert "[rowsplit]" not in whole["_stderr"] and split["_stderr"].count("[rowsplit]") > 100, split["_stderr"][-2000:]
    live = whole["mask"] != 0
    assert np.abs(split["encoder_out"] - whole["encoder_out"])[live].max() < 2e-5
    for k in ("auto", "fork3", "fork5_7"):
        assert (split[k + "_tokens"] == whole[k + "_tokens"]).all(), k
        assert np.abs(split[k + "_scores"] - whole[k + "_scores"]).max() < 1e-5, k
        assert (split[k + "_forks"] == whole[k + "_forks"]).all(), k
