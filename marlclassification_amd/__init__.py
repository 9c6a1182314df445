"""marlclassification_amd: MI355X-native hot path of MARLClassification (multi-agent episode
rollout + A2C update) behind the reference's Python surface.  All arithmetic lives in
csrc/libmarl_hip.so (hand-written HIP for gfx950), bound through ctypes (_lib.py)."""

__all__ = ["core", "networks", "training", "engine", "fused", "parallel", "metrics"]
