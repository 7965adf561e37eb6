"""``chatglm_q_amd.decoder.SentencePieceIds``: the token ids ``ChatGLMDecoder.from_pretrained`` builds when the checkpoint folder holds a
sentencepiece file, against the behaviour of the reference's tokenizer (chatglm_q/tokenizer.py:24-64, restated here as expectations: the
five special tokens sit behind the sentencepiece vocabulary in the order [MASK] [gMASK] [sMASK] <sop> <eop>; ``encode`` prepends [gMASK]
<sop>; ``decode`` drops ids past the sentencepiece vocabulary; ``tok[name]`` resolves specials and pieces).  The model is trained here on a
few lines (sentencepiece's own trainer): no file from the reference is involved.  Also: ``save_pretrained`` writes the tokenizer file back."""
import os

import pytest
import torch

sentencepiece = pytest.importorskip("sentencepiece")

from chatglm_q_amd.decoder import ChatGLMDecoder, SentencePieceIds  # noqa: E402


@pytest.fixture(scope="module")
def sp_model(tmp_path_factory):
    d = tmp_path_factory.mktemp("sp")
    corpus = d / "corpus.txt"
    corpus.write_text("\n".join([f"the quick brown fox jumps over the lazy dog {i}" for i in range(200)] +
                                ["hello world this is a tiny corpus for a tiny model"] * 50))
    sentencepiece.SentencePieceTrainer.train(input=str(corpus), model_prefix=str(d / "sentencepiece"), vocab_size=60, model_type="bpe",
                                             minloglevel=2)
    return d / "sentencepiece.model"


def test_ids_follow_the_reference_tokenizer_layout(sp_model):
    tok = SentencePieceIds(sp_model)
    sp = sentencepiece.SentencePieceProcessor(model_file=str(sp_model))
    n = len(sp)
    assert len(tok) == n + 5
    assert [tok[t] for t in ("[MASK]", "[gMASK]", "[sMASK]", "<sop>", "<eop>")] == [n, n + 1, n + 2, n + 3, n + 4]
    assert tok["</s>"] == sp.piece_to_id("</s>") == sp.eos_id()
    with pytest.raises(KeyError):
        tok["no such piece at all"]
    ids = tok.encode("hello world")
    assert ids[:2] == [n + 1, n + 3] and ids[2:] == sp.encode("hello world")
    assert tok.encode("hello world", add_special_tokens=False) == sp.encode("hello world")
    # decode drops everything past the sentencepiece vocabulary (the special ids would make sentencepiece raise)
    assert tok.decode(ids + [n + 4, n + 17]) == "hello world"
    assert tok.decode(torch.tensor(ids)) == "hello world"


def test_from_pretrained_builds_it_and_save_pretrained_writes_the_file_back(sp_model, tmp_path):
    from chatglm_q_amd import loader as L
    from chatglm_q_amd import model as M
    cfg = M.ChatGLM2Config(hidden_size=64, inner_hidden_size=96, head_hidden_size=16, num_multi_query_groups=2, num_attention_heads=4,
                           num_layers=1, vocab_size=96, max_sequence_length=32)
    lc = L.ChatGLMLoadConfig(model_config=cfg, quant_type="int4g32", torch_dtype="float32")
    model = L.build_model(lc)
    with torch.no_grad():
        for k, v in model.state_dict().items():
            v.copy_(torch.randint(0, 255, v.shape, dtype=torch.uint8) if v.dtype == torch.uint8 else torch.rand(v.shape) * 0.05)
    L.save_model(tmp_path / "a", lc, model, tokenizer_file=sp_model)
    dec = ChatGLMDecoder.from_pretrained(tmp_path / "a", device="cpu")
    assert isinstance(dec.tokenizer, SentencePieceIds) and dec.eos_token_id == dec.tokenizer["</s>"]
    assert os.path.samefile(dec.tokenizer_file, tmp_path / "a" / lc.tokenizer_file)
    dec.save_pretrained(tmp_path / "b")
    assert (tmp_path / "b" / lc.tokenizer_file).read_bytes() == sp_model.read_bytes()
    again = ChatGLMDecoder.from_pretrained(tmp_path / "b", device="cpu")
    assert again.tokenizer.encode("hello world") == dec.tokenizer.encode("hello world")
    for k, v in again.model.state_dict().items():
        assert torch.equal(v, dec.model.state_dict()[k]), k
