from primitive3d_amd import __version__  # noqa: F401
