"""`slowfast` namespace of the reference, served by csts_amd: ``slowfast.models.build_model`` and the modules the
CSTS path touches resolve to the MI355X implementation, so reference-side callers keep their imports."""
