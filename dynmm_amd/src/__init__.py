"""Reference-named import surface: `from dynmm_amd.src.X import Y` mirrors FusionDynMM's
`from src.X import Y` for the hot path.  Implementations live in dynmm_amd.nn."""
