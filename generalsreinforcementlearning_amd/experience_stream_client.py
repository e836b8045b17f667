"""python/experience_stream_client.py under its own module and class names, for code written against it
(rl_training_example.py:14: `from experience_stream_client import ExperienceStreamClient, ExperienceConfig,
ExperienceDataset`): the same three names, `ExperienceStreamClient(config)` with the reference's constructor - the
experiences come from a local engine on the GPU (experience_stream.configure_local_engine) instead of a gRPC server."""
from .experience_stream import ExperienceConfig, ExperienceDataset, ExperienceStreamClient, configure_local_engine

__all__ = ["ExperienceStreamClient", "ExperienceConfig", "ExperienceDataset", "configure_local_engine"]
