"""Beam search (quickvideo_amd/beam.py) pinned against the installed transformers: the same tiny random text model decoded by HF
`generate(num_beams=...)` and by beam.beam_search driven through a cache-less `advance` callback over that model — token sequences must be
identical for several beam widths, length penalties, eos placements, early stopping modes and a repetition penalty."""
import pytest
import torch

from quickvideo_amd.beam import beam_search


def _tiny_model(vocab=97, seed=0):
    from transformers import Qwen2Config, Qwen2ForCausalLM
    torch.manual_seed(seed)
    cfg = Qwen2Config(vocab_size=vocab, hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                      max_position_embeddings=128, tie_word_embeddings=False)
    m = Qwen2ForCausalLM(cfg).eval()
    with torch.no_grad():                                        # spread the logits: random init is nearly uniform, beams would tie
        for p in m.parameters():
            p.mul_(6.0)
    return m


@pytest.mark.parametrize("B,lp,eos,es,rp,mnt", [
    (2, 1.0, (), False, 1.0, 6), (3, 1.0, (5,), False, 1.0, 8), (4, 1.0, (5, 9), False, 1.0, 8), (3, 2.0, (5,), False, 1.0, 8),
    (3, 0.0, (5,), False, 1.0, 8), (3, 1.0, (5,), True, 1.0, 10), (4, 1.0, (7,), "never", 1.0, 8), (3, 1.0, (5,), False, 1.3, 8),
    (1, 1.0, (5,), False, 1.0, 6), (5, 1.0, (3, 4, 5, 6, 7, 8, 9, 10, 11, 12), False, 1.0, 12),
])
def test_beam_search_equals_hf_generate(B, lp, eos, es, rp, mnt):
    m = _tiny_model()
    for seed in range(4):
        g = torch.Generator().manual_seed(100 + seed)
        prompt = torch.randint(13, 97, (1, 7), generator=g)
        kw = dict(num_beams=B, max_new_tokens=mnt, do_sample=False, length_penalty=lp, early_stopping=es, repetition_penalty=rp,
                  eos_token_id=list(eos) if eos else None, pad_token_id=0, num_return_sequences=1)
        with torch.no_grad():
            want = m.generate(prompt, **kw)[0, prompt.shape[1]:].tolist()
        while want and want[-1] == 0 and (not eos or 0 not in eos):                # HF pads a sequence that ended early
            want.pop()
        seqs = [prompt[0].tolist() for _ in range(B)]

        def advance(parents, tokens):
            nonlocal seqs
            seqs = [seqs[p] + [t] for p, t in zip(parents, tokens)]
            with torch.no_grad():
                return torch.stack([m(torch.tensor([s])).logits[0, -1].float() for s in seqs])

        with torch.no_grad():
            first = m(prompt).logits[0, -1].float()
        got = beam_search(first, advance, B, mnt, eos_ids=eos, length_penalty=lp, early_stopping=es, repetition_penalty=rp, prompt_ids=prompt[0].tolist())
        assert got == want, (seed, got, want)


@pytest.mark.parametrize("B,kw", [(3, dict(temperature=1.3)), (4, dict(top_k=12)), (3, dict(top_p=0.8, temperature=0.9)), (2, dict())])
def test_beam_sampling_equals_hf_generate(B, kw):
    """num_beams > 1 with do_sample=True (beam-search multinomial sampling): the same draws from the same RNG state give HF's sequences."""
    m = _tiny_model()
    for seed in range(3):
        g = torch.Generator().manual_seed(200 + seed)
        prompt = torch.randint(13, 97, (1, 7), generator=g)
        torch.manual_seed(77 + seed)
        with torch.no_grad():
            want = m.generate(prompt, num_beams=B, max_new_tokens=7, do_sample=True, eos_token_id=[5], pad_token_id=0, **kw)[0, 7:].tolist()
        while want and want[-1] == 0:
            want.pop()
        seqs = [prompt[0].tolist() for _ in range(B)]

        def advance(parents, tokens):
            nonlocal seqs
            seqs = [seqs[p] + [t] for p, t in zip(parents, tokens)]
            with torch.no_grad():
                return torch.stack([m(torch.tensor([s])).logits[0, -1].float() for s in seqs])

        with torch.no_grad():
            first = m(prompt).logits[0, -1].float()
        torch.manual_seed(77 + seed)
        got = beam_search(first, advance, B, 7, eos_ids=(5,), do_sample=True, temperature=kw.get("temperature", 1.0), top_k=kw.get("top_k", 0),
                          top_p=kw.get("top_p", 1.0))
        assert got == want, (seed, got, want)


def test_beams_actually_differ_from_greedy_somewhere():
    """The comparison above must not be vacuous: on this model some prompt makes the 4-beam answer differ from the greedy one."""
    m = _tiny_model()
    diff = 0
    for seed in range(8):
        g = torch.Generator().manual_seed(100 + seed)
        prompt = torch.randint(13, 97, (1, 7), generator=g)
        with torch.no_grad():
            a = m.generate(prompt, num_beams=4, max_new_tokens=8, do_sample=False, pad_token_id=0, eos_token_id=None)[0].tolist()
            b = m.generate(prompt, num_beams=1, max_new_tokens=8, do_sample=False, pad_token_id=0, eos_token_id=None)[0].tolist()
        diff += a != b
    assert diff > 0


def test_engine_beams_share_the_prefilled_cache_and_equal_a_from_scratch_search():
    """EngineBeams (one KV arena: the prefilled rows shared, per-beam tails swapped in) against the same search whose `advance` rebuilds
    every beam FROM SCRATCH (fresh engine: all groups, prompt tail, then the beam's tokens one by one).  Same token sequence; the arena's
    shared rows are untouched; afterwards the engine is back at the prefill state."""
    import numpy as np
    from oracle_ops import OracleOps
    from oracle import qp_oracle as O
    from quickvideo_amd.beam import EngineBeams
    from quickvideo_amd.engine import QuickPrefillEngine
    from quickvideo_amd.lvu_config import LVUConfig
    from quickvideo_amd.spec import TextSpec
    from quickvideo_amd.weights import DecoderWeights
    dims = dict(hidden=256, n_heads=2, n_kv_heads=1, head_dim=128, intermediate=256, n_layers=2, vocab=64)
    so, spec = O.TextSpec(**dims), TextSpec(**dims)
    w = {k: (v * (8.0 if "lm_head" in k else 1.0)).to(torch.bfloat16) for k, v in O.synthetic_text_weights(so, seed=13, norm_jitter=0.1).items()}
    rs = np.random.RandomState(3)
    groups, tail = [20, 24], 6
    T = sum(groups) + tail
    embeds = torch.from_numpy(rs.standard_normal((T, 256)).astype(np.float32) * 0.5).to(torch.bfloat16)
    pos = torch.from_numpy(np.tile(np.arange(T, dtype=np.int64), (3, 1)))
    cfg = LVUConfig("x", top_p=0.5, video_group_size=4)
    dw = DecoderWeights.from_named(spec, w, "cpu")

    def prefill():
        eng = QuickPrefillEngine(dw, cfg, capacity=T + 16, max_group_tokens=max(groups), device="cpu", ops=OracleOps())
        st = 0
        for n in groups:
            eng.prefill_group(embeds[st:st + n], pos[:, st:st + n]); st += n
        return eng, eng.prefill_tail(embeds[st:], pos[:, st:])

    B, mnt = 3, 5
    eng, first = prefill()
    shared = [eng.arena.buf[l, :, :, :eng.arena.len[l]].clone() for l in range(2)]
    lens = list(eng.arena.len)
    beams = EngineBeams(eng, 0, B, mnt)
    got = beam_search(first, beams.advance, B, mnt, eos_ids=(7,))
    beams.finish(len(got))
    assert list(eng.arena.len) == lens and all(torch.equal(eng.arena.buf[l, :, :, :lens[l]], shared[l]) for l in range(2))

    hist = [[] for _ in range(B)]

    def advance_scratch(parents, tokens):
        nonlocal hist
        hist = [hist[p] + [t] for p, t in zip(parents, tokens)]
        out = []
        for h in hist:
            e2, _ = prefill()
            for tok in h:
                lg = e2.decode_step(e2.embed_tokens(torch.tensor([tok])), 0)
            out.append(lg.float())
        return torch.stack(out)

    want = beam_search(first, advance_scratch, B, mnt, eos_ids=(7,))
    assert got == want and 1 <= len(got) <= mnt
