from .trainer import MetricLogger, Trainer
