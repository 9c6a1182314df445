"""Run configuration objects and the ``marl.json`` wire format of the reference
(config.py:11-131): same field names, same JSON keys, so a ``marl.json`` written by either
side loads on the other.  ``build_marl`` is the constructor path for the hot-path objects."""

import json
from os.path import exists, isfile
from typing import List, Tuple

from pydantic import BaseModel

from .core import Environment, MultiAgent
from .networks import ModelsWrapper
from .networks.vision import CNN_BY_NAME

_MODEL_KEYS = (
    "ft_extr_str", "window_size", "hidden_size_belief", "hidden_size_action", "hidden_size_msg",
    "hidden_size_msg_output", "hidden_size_state", "state_dim", "actions", "nb_class",
    "hidden_size_linear_belief", "hidden_size_linear_action",
)


class MainConfig(BaseModel):
    step: int
    run_id: str
    cuda: bool
    nb_agent: int


class ModelConfig(BaseModel):
    ft_extr_str: str
    window_size: int
    hidden_size_belief: int
    hidden_size_action: int
    hidden_size_msg: int
    hidden_size_msg_output: int
    hidden_size_state: int
    state_dim: int
    actions: List[List[int]]
    nb_class: int
    hidden_size_linear_belief: int
    hidden_size_linear_action: int

    def save_marl_config(self, out_json_path: str) -> None:
        with open(out_json_path, "w", encoding="utf-8") as f:
            json.dump({k: getattr(self, k) for k in _MODEL_KEYS}, f)

    @classmethod
    def load_marl_config(cls, json_path: str) -> "ModelConfig":
        assert exists(json_path) and isfile(json_path), f'"{json_path}" does not exist or is not a file'
        with open(json_path, "r", encoding="utf-8") as f:
            raw = json.load(f)
        return cls(**{k: raw[k] for k in _MODEL_KEYS})

    def build_networks(self) -> ModelsWrapper:
        assert self.ft_extr_str in CNN_BY_NAME, (
            f'Unknown feature extractor "{self.ft_extr_str}", expected one of {sorted(CNN_BY_NAME)}'
        )
        return ModelsWrapper(
            CNN_BY_NAME[self.ft_extr_str](self.window_size),
            self.hidden_size_belief, self.hidden_size_action, self.hidden_size_msg,
            self.hidden_size_msg_output, self.hidden_size_state, self.state_dim,
            len(self.actions), self.nb_class, self.hidden_size_linear_belief,
            self.hidden_size_linear_action,
        )

    def build_environment(self) -> Environment:
        return Environment(self.actions, self.window_size)

    def build_marl(self, nb_agents: int) -> Tuple[ModelsWrapper, MultiAgent, Environment]:
        networks = self.build_networks()
        return networks, MultiAgent(nb_agents, networks), self.build_environment()


class TrainConfig(BaseModel):
    img_size: int
    nb_epoch: int
    learning_rate: float
    batch_size: int
    resources_dir: str
    output_dir: str
    gamma: float


class EvalConfig(BaseModel):
    img_size: int
    state_dict_path: str
    batch_size: int
    json_path: str
    dataset_path: str
    output_dir: str


class InferConfig(BaseModel):
    state_dict_path: str
    json_path: str
    images_path: List[str]
    output_dir: str
    class_to_idx: str
