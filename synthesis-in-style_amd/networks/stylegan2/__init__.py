"""Same exports as the reference package (networks/stylegan2/__init__.py:1)."""
from networks.stylegan2.model import Discriminator, Generator  # noqa: F401
