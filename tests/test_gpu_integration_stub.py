"""-m gpu: the ctypes stub INTEGRATION.md §B hands to a maintainer of the reference is executed AS IT STANDS IN THE DOCUMENT
(the first python block of §B; only the library path is made absolute): its own rpr_ctx, raw model / trie handles, the status
words, the two repeat rules. Its outputs are compared with the golden (= the reference's own run) and with the package's
boundary function on the same inputs."""
import os
import re

import numpy as np
import pytest
import torch

from conftest import compare_ranked

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub_namespace():
    text = open(os.path.join(REPO, "INTEGRATION.md")).read()
    sec = text[text.index("## B. "):]
    code = re.search(r"```python\n(.*?)```", sec, re.S).group(1)
    assert 'C.CDLL("libripor_hip.so")' in code
    code = code.replace('C.CDLL("libripor_hip.so")', 'C.CDLL(%r)' % os.path.join(REPO, "ripor_amd", "libripor_hip.so"))
    ns = {}
    exec(compile(code, "INTEGRATION.md#B", "exec"), ns)
    return ns


def test_the_stub_of_the_integration_guide_runs_and_matches(golden_cache):
    from ripor_amd import engine as E
    ns = _stub_namespace()
    for name, log_softmax in (("g1_mini_b10_l32", False), ("g2_base_b10_l32", False)):
        g = golden_cache(name)
        if g.log_softmax:
            continue
        # model and trie handles belong to the ctx they were loaded under: the stub's own ctx, wrapped for the package's packers
        # (what `rpr_load_model` / `rpr_build_trie` want is in INTEGRATION.md's table; the packers are not the subject here)
        sctx = object.__new__(E.Context)
        sctx.lib, sctx.device, sctx.handle = E._lib.load(), torch.device("cuda", 0), ns["_ctx"]
        model = E.DeviceModel(sctx, g.state_dict, g.dims)
        trie = E.DeviceTrie.from_codes(sctx, g.codes, g.V)
        ids, mask = torch.from_numpy(g.input_ids).cuda(), torch.from_numpy(g.attention_mask).cuda()
        seqs, scores = ns["hip_generate"](model.handle, trie.handle, ids, mask, g.L, g.B)
        torch.cuda.synchronize()
        assert seqs.shape == (g.Q * g.B, g.L + 1) and scores.shape == (g.Q * g.B,) and seqs.dtype == torch.long
        assert (seqs[:, 0] == 0).all()
        tok = seqs.view(g.Q, g.B, g.L + 1)[:, :, 1:].cpu().numpy()
        compare_ranked(g, tok, scores.view(g.Q, g.B).cpu().numpy(), label=" (INTEGRATION.md stub)")
        ctx = E.Context.get(0)                           # the package's boundary on its own ctx, model and trie
        model2, trie2 = E.DeviceModel(ctx, g.state_dict, g.dims), E.DeviceTrie.from_codes(ctx, g.codes, g.V)
        ref = E.search(model2, trie2, ids.cpu(), mask.cpu(), g.B, g.L)
        torch.cuda.synchronize()
        assert (ref.tokens.cpu().numpy() == tok).all() and torch.equal(ref.scores.cpu().view(-1), scores.cpu())
    # the error path of the stub: an all-zero attention row is reported, a bad argument raises with the library's message
    bad = mask.clone(); bad[0] = 0
    with pytest.raises(ValueError, match="all zero"):
        ns["hip_generate"](model.handle, trie.handle, ids, bad, g.L, g.B)
    with pytest.raises(RuntimeError, match="exceeds the model.s decoder length"):
        ns["hip_generate"](model.handle, trie.handle, ids, mask, g.L + 100, g.B)
