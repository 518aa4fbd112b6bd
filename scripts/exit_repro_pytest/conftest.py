import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from conftest import *  # noqa: F401,F403  (the suite's own fixtures: oracle, golden, spd ...)
