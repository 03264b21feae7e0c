from neurst_amd.layers.search.beam_search import BeamSearch, sequence_beam_search  # noqa: F401
