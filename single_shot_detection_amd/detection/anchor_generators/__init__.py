from . import retina_net, ssd  # noqa: F401  (detection/anchor_generators/__init__.py:1)
