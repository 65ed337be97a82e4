"""orbx: MI355X-native ORB front-end (extractor, matcher primitives, bag of words) behind the reference's
ORBextractor / ORBmatcher / ORBVocabulary interfaces.  All compute lives in liborbx.so (HIP, gfx950)."""
from .extractor import ORBextractor  # noqa: F401
from .matcher import ORBmatcher  # noqa: F401
from .vocabulary import ORBVocabulary  # noqa: F401
from .kfdb import KeyFrameDatabase  # noqa: F401
from ._lib import KP_DTYPE, OrbxError  # noqa: F401
