from .optim import FusedAdamW
from .distributed import DataParallel, init_process_group_from_env, allreduce_gradients
