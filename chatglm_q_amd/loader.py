"""Checkpoint format I/O compatible with K024/chatglm-q model folders.

Folder layout (chatglm_q/loader.py:16-38,69-159): ``config.json`` = the load config below, one or more
``*.safetensors`` shards whose keys are the model's state_dict keys, plus a tokenizer file that this build does
not interpret.  Loading copies each tensor straight into ``model.state_dict()`` buffers, exactly as the reference
does (chatglm_q/loader.py:90-104); the quantized modules notice the in-place refill through the buffers' version
counters and rebuild their derived decode layout on the next forward.
"""
from __future__ import annotations

import json
import os
from dataclasses import asdict, dataclass, field
from pathlib import Path
from typing import Optional, Union

import torch

from .model import ChatGLM2Config, ChatGLM2Model, create_quant_int4_model, create_quant_int8_model


@dataclass
class ChatGLMLoadConfig:
    # same fields, defaults and JSON form as chatglm_q/loader.py:16-38
    model_type: str = "ChatGLM2Model"
    model_config: ChatGLM2Config = field(default_factory=ChatGLM2Config)
    quant_type: str = "none"                     # "none" | "int8" | "int4g32"
    weight_files: list = field(default_factory=list)
    tokenizer_file: str = "sentencepiece.model"
    torch_dtype: str = "float32"                 # "float32" | "float16" | "bfloat16"

    def __post_init__(self):
        if self.model_type != "ChatGLM2Model":
            raise AssertionError("Only 'ChatGLM2Model' is supported")
        if not isinstance(self.model_config, ChatGLM2Config):
            self.model_config = ChatGLM2Config(**self.model_config)
        if self.quant_type not in ("none", "int8", "int4g32"):
            raise AssertionError(f"No quant_type named '{self.quant_type}'")

    def get_torch_dtype(self):
        return getattr(torch, self.torch_dtype)

    @staticmethod
    def from_json(json_str):
        return ChatGLMLoadConfig(**json.loads(json_str))

    def to_json(self):
        return json.dumps(asdict(self), ensure_ascii=False, indent=2)


def build_model(config: ChatGLMLoadConfig, torch_dtype=None) -> ChatGLM2Model:
    dtype = torch_dtype or config.get_torch_dtype()
    if config.quant_type == "none":
        return ChatGLM2Model(config.model_config, dtype)
    if config.quant_type == "int8":
        return create_quant_int8_model(config.model_config, dtype)
    return create_quant_int4_model(config.model_config, 32, dtype)


@torch.no_grad()
def load_model(model_path: Union[str, Path], torch_dtype=None, device=None):
    """Returns (load_config, model).  Extra keys in the files are reported and ignored, keys the files do not
    provide are reported as uninitialised (chatglm_q/loader.py:97-99,109-110)."""
    from safetensors import safe_open
    model_path = Path(model_path)
    config = ChatGLMLoadConfig.from_json((model_path / "config.json").read_text())
    model = build_model(config, torch_dtype)
    state = dict(model.state_dict())
    for name in config.weight_files:
        with safe_open(model_path / name, framework="pt") as f:
            for key in f.keys():
                if key not in state:
                    print(f'"{key}" is ignored')
                    continue
                value = f.get_tensor(key)
                if state[key].is_floating_point():
                    value = value.type_as(state[key])
                state[key].copy_(value.to(state[key].device))
                state.pop(key)
    if state:
        print(f'model weights "{", ".join(state.keys())}" are not initialized')
    if device is not None:
        model.to(device=device)
    return config, model


def shard_plan(state_dict, max_shard_bytes: int = 2 * 1024 ** 3):
    """Greedy packing in state_dict order into files of at most max_shard_bytes (chatglm_q/loader.py:139-150)."""
    mapping, index, size = {}, 0, 0
    for name, weight in state_dict.items():
        nbytes = weight.element_size() * weight.numel()
        if size + nbytes > max_shard_bytes:
            index += 1
            size = 0
        size += nbytes
        mapping[name] = f"model_weights_{index}.safetensors"
    return mapping


def save_model(path: Union[str, Path], config: ChatGLMLoadConfig, model: ChatGLM2Model, shard: bool = True,
               max_shard_bytes: int = 2 * 1024 ** 3, tokenizer_file: Optional[Union[str, Path]] = None):
    from safetensors.torch import save_file
    import shutil
    path = Path(path)
    path.mkdir(parents=True, exist_ok=True)
    if tokenizer_file is not None:
        dst = path / config.tokenizer_file
        if not (dst.exists() and os.path.samefile(tokenizer_file, dst)):      # saving back into the folder it was loaded from
            shutil.copy(tokenizer_file, dst)
    state = {k: v.detach().cpu().contiguous() for k, v in model.state_dict().items()}
    if not shard:
        config.weight_files = ["model_weights.safetensors"]
        save_file(state, str(path / config.weight_files[0]))
    else:
        mapping = shard_plan(state, max_shard_bytes)
        config.weight_files = sorted(set(mapping.values()))
        for name in config.weight_files:
            save_file({k: state[k] for k, f in mapping.items() if f == name}, str(path / name))
    (path / "config.json").write_text(config.to_json())
