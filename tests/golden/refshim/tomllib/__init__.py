from tomli import *  # noqa: F401,F403  (py3.10 has no tomllib)
from tomli import load, loads  # noqa: F401
