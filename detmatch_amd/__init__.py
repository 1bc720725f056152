"""detmatch_amd — MI355X-native DetMatch training step."""
import os

_MIOPEN_DB = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'miopen_db')


def _miopen_env():
    """MIOpen reads its environment when it initialises (first convolution of the process), so the
    variables are set when the package is imported, not when a workload asks for the tuned db."""
    if os.path.isdir(_MIOPEN_DB) and os.access(_MIOPEN_DB, os.W_OK) and \
            any(f.endswith('.ufdb.txt') for f in os.listdir(_MIOPEN_DB)):
        os.environ.setdefault('MIOPEN_USER_DB_PATH', _MIOPEN_DB)
        os.environ.setdefault('MIOPEN_FIND_MODE', 'FAST')


_miopen_env()


def enable_tuned_miopen():
    """Dense convolutions (BEV backbone, ResNet-50/FPN) run in MIOpen.  `miopen_db/` holds the
    find-db MIOpen produced ONCE for every convolution shape of the step (tools/miopen_tune.sh,
    ~10 min of exhaustive search); with MIOPEN_FIND_MODE=FAST a run only LOOKS the best solver up
    (unknown shapes fall back to immediate mode), so no run pays the search.  Must be called before
    the first convolution.  Returns True when the tuned db is in use."""
    import torch
    if not os.path.isdir(_MIOPEN_DB) or not any(f.endswith('.ufdb.txt') for f in os.listdir(_MIOPEN_DB)):
        return False
    if not os.access(_MIOPEN_DB, os.W_OK):       # MIOpen opens its user db read-write
        return False
    os.environ.setdefault('MIOPEN_USER_DB_PATH', _MIOPEN_DB)
    os.environ.setdefault('MIOPEN_FIND_MODE', 'FAST')
    if os.environ['MIOPEN_USER_DB_PATH'] != _MIOPEN_DB:
        return False
    torch.backends.cudnn.benchmark = True
    return True
