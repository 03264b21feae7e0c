from neurst_amd.layers.search.beam_search import (BeamSearch, SequenceSearch, build_search_layer,  # noqa: F401
                                                  sequence_beam_search)
from neurst_amd.layers.search import sampling  # noqa: E402,F401
