from .agent import AgentOutput, MultiAgent
from .environment import Environment
from .episode import EpisodeDetailedOutput, EpisodeOutput, EpisodeSampler
