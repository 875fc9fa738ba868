"""CPU experiment (oracle only, not product code): logit / beam-score error of the search when the self-attention K/V
cache is stored as f16 hi + int8 lo (3 bytes per element, 19-20 significant bits) instead of fp32.
Usage: python tools/kv_format_probe.py [fixture]"""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import conftest
from oracle import beam_ref, t5_ref
from ripor_amd.utils import synth


def q3(x: torch.Tensor) -> torch.Tensor:
    """x -> hi (f16, round to nearest) + q * 2^(e-18), q int8 in [-127, 127], e = unbiased exponent of hi."""
    hi = x.to(torch.float16)
    hif = hi.to(torch.float32)
    r = x - hif
    _, ex = torch.frexp(hif)                       # hif = m * 2^ex, m in [0.5, 1)  ->  ulp(hi) = 2^(ex - 11)
    ex = torch.clamp(ex, min=-13)                  # subnormal halves share the smallest normal exponent
    step = torch.ldexp(torch.ones_like(x), ex - 12 - 7)   # half an ulp spread over 127 levels ~ 2^(ex-19)
    q = torch.clamp(torch.round(r / step), -127, 127)
    return hif + q * step


def q3row(x: torch.Tensor) -> torch.Tensor:
    """Row-scaled variant: hi = f16(x); lo = int8 of (x - hi) in units of s = 2^(Emax - 18), Emax = exponent of the
    largest |hi| of the 64-element head row (last dim)."""
    hi = x.to(torch.float16).to(torch.float32)
    r = x - hi
    _, ex = torch.frexp(hi.abs().amax(dim=-1, keepdim=True))     # max = m * 2^ex, m in [0.5, 1)
    step = torch.ldexp(torch.ones_like(r[..., :1]), ex - 1 - 18)
    q = torch.clamp(torch.round(r / step), -127, 127)
    return hi + q * step


MODE = q3


class Q3Cached(t5_ref.T5RefCached):
    def step(self, last_tokens, R):
        out = super().step(last_tokens, R)
        for i in range(len(self.k_cache)):        # requantise only the newest position (earlier ones are already q3)
            self.k_cache[i][:, :, -1:] = MODE(self.k_cache[i][:, :, -1:])
            self.v_cache[i][:, :, -1:] = MODE(self.v_cache[i][:, :, -1:])
        return out


name = sys.argv[1] if len(sys.argv) > 1 else "g2_base_b10_l32"
if len(sys.argv) > 2 and sys.argv[2] == "row":
    MODE = q3row
g = conftest.Golden(name)
pm = beam_ref.PrefixMaskRef(beam_ref.build_list_smtid_to_nextids(synth.codes_to_docid_to_smtid(g.codes)), g.V)
torch.set_num_threads(8)
ra, rb = {}, {}
sa, sca = beam_ref.beam_search_ref(t5_ref.T5RefCached(g.state_dict, g.dims), pm, g.input_ids, g.attention_mask, g.B, g.L,
                                   use_kv_cache=True, record=ra)
sb, scb = beam_ref.beam_search_ref(Q3Cached(g.state_dict, g.dims), pm, g.input_ids, g.attention_mask, g.B, g.L,
                                   use_kv_cache=True, record=rb)
x = torch.randn(100000) * 3
print("q3 relative error max", float(((q3(x) - x).abs() / x.abs().clamp_min(1e-3)).max()))
print("sequences identical:", bool((sa == sb).all()), " max beam-score diff", float((sca - scb).abs().max()))
errs = [float(np.abs(ra["steps"][t]["logits"] - rb["steps"][t]["logits"]).max()) for t in range(g.L)
        if (ra["steps"][t]["top_tok"] == rb["steps"][t]["top_tok"]).all()]
print("max logit diff per step (steps with identical selections):", ["%.1e" % e for e in errs])
