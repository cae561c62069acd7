from .env_L2A import EnvMaxcut  # noqa: F401
