"""Same exports as the reference package (networks/swagan/__init__.py:1)."""
from networks.swagan.model import Discriminator, Generator  # noqa: F401
