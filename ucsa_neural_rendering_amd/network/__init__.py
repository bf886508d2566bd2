from .deeplabv3 import *  # noqa: F401,F403  (reference nr4seg/network/__init__.py)
