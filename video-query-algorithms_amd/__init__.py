"""MI355X-native hot path of PARC-projects/video-query-algorithms.

Hot path B (similarity scoring / ranking / weight update): :class:`FeatureDB`,
:class:`TicketScoring` / :class:`Ticket`, :class:`TargetClip`, :class:`Hyperparameter`,
:func:`compute_matches`.  Hot path A (TSN BN-Inception feature extraction): ``tsn`` subpackage.
Everything computes through ``libvqamd.so`` (include/vq_amd.h); there is no CPU fallback.
"""
from ._lib import VqError, load as load_library          # noqa: F401
from .feature_db import FeatureDB                         # noqa: F401
from .ticket import Ticket, TicketScoring, ScoreMap, SimilarityMap, install   # noqa: F401
from .target_clip import TargetClip                       # noqa: F401
from .hyperparameter import Hyperparameter                # noqa: F401
from .compute_matches import compute_matches              # noqa: F401
