"""Checkpoint folder format (config.json + safetensors shards) against the reference's own output."""
import json
import os

import torch

from chatglm_q_amd import loader as L
from chatglm_q_amd import model as M

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "loader_format.json")))


def tiny_cfg():
    return M.ChatGLM2Config(hidden_size=128, inner_hidden_size=224, head_hidden_size=32, num_multi_query_groups=2,
                            num_attention_heads=4, num_layers=2, vocab_size=256, max_sequence_length=64)


def test_config_json_text_and_shard_plan_match_reference():
    lc = L.ChatGLMLoadConfig(model_config=tiny_cfg(), quant_type="int4g32",
                             weight_files=["model_weights_0.safetensors", "model_weights_1.safetensors"], torch_dtype="float16")
    assert lc.to_json() == GOLD["config_json"]
    back = L.ChatGLMLoadConfig.from_json(GOLD["config_json"])
    assert back.model_config == tiny_cfg() and back.quant_type == "int4g32" and back.get_torch_dtype() == torch.float16
    model = L.build_model(lc)
    assert list(model.state_dict().keys()) == GOLD["state_keys"]
    assert L.shard_plan(model.state_dict(), 40000) == GOLD["shard_mapping_40000"]


def test_save_load_roundtrip(tmp_path):
    lc = L.ChatGLMLoadConfig(model_config=tiny_cfg(), quant_type="int4g32", torch_dtype="float32")
    model = M.fill_synthetic_(L.build_model(lc), seed=3)
    L.save_model(tmp_path, lc, model, shard=True, max_shard_bytes=40000)
    assert len(lc.weight_files) > 1 and (tmp_path / "config.json").exists()
    lc2, model2 = L.load_model(tmp_path)
    assert lc2.weight_files == lc.weight_files
    for (k1, v1), (k2, v2) in zip(model.state_dict().items(), model2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)
    ids = torch.randint(0, 256, (1, 6))
    with torch.no_grad():
        assert torch.equal(model(input_ids=ids)[1], model2(input_ids=ids)[1])


def test_decoder_from_pretrained_folder_roundtrip(tmp_path):
    """ChatGLMDecoder.from_pretrained / save_pretrained (chatglm_q/decoder.py:49-61) on a checkpoint folder: same greedy ids as the
    model the folder was written from; a missing folder is an error (hub ids are not resolved)."""
    import pytest
    from chatglm_q_amd.decoder import ChatGLMDecoder
    lc = L.ChatGLMLoadConfig(model_config=tiny_cfg(), quant_type="int4g32", torch_dtype="float32")
    model = M.fill_synthetic_(L.build_model(lc), seed=5)
    L.save_model(tmp_path / "m", lc, model, shard=True, max_shard_bytes=40000)
    dec = ChatGLMDecoder.from_pretrained(tmp_path / "m", device="cpu")
    assert dec.tokenizer is None and dec.max_sequence_length == 64
    ref = ChatGLMDecoder(lc, model.eval())
    a = list(dec.generate_ids([3, 7, 11], max_generated_tokens=5, greedy=True, ignore_eos=True))
    b = list(ref.generate_ids([3, 7, 11], max_generated_tokens=5, greedy=True, ignore_eos=True))
    assert a == b and len(a) == 5
    dec.save_pretrained(tmp_path / "again")
    again = L.ChatGLMLoadConfig.from_json((tmp_path / "again" / "config.json").read_text())
    assert again.model_config == lc.model_config and again.quant_type == "int4g32" and again.weight_files
    with pytest.raises(FileNotFoundError):
        ChatGLMDecoder.from_pretrained(tmp_path / "nope")


def test_save_back_into_the_folder_it_was_loaded_from(tmp_path):
    """from_pretrained records the folder's tokenizer file and save_pretrained copies it - saving back into the same folder
    raised shutil.SameFileError.  And a corrupt sentencepiece file no longer aborts from_pretrained (token ids are still served)."""
    import warnings
    from chatglm_q_amd.decoder import ChatGLMDecoder
    lc = L.ChatGLMLoadConfig(model_config=tiny_cfg(), quant_type="int4g32", torch_dtype="float32")
    model = M.fill_synthetic_(L.build_model(lc), seed=6)
    folder = tmp_path / "m"
    L.save_model(folder, lc, model, shard=False)
    (folder / lc.tokenizer_file).write_bytes(b"not a sentencepiece model")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        dec = ChatGLMDecoder.from_pretrained(folder, device="cpu")
    assert dec.tokenizer is None and any("no tokenizer built" in str(x.message) for x in w)
    assert dec.tokenizer_file is not None
    dec.save_pretrained(folder)                                                # same folder: no SameFileError
    assert (folder / lc.tokenizer_file).read_bytes() == b"not a sentencepiece model"
    dec.save_pretrained(tmp_path / "other")
    assert (tmp_path / "other" / lc.tokenizer_file).read_bytes() == b"not a sentencepiece model"
