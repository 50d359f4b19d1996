from .objectives import *

from .cqtwrapper import CQT
from .modules import *
