"""Command-line entry points: train, exporter, semantic_projection, depth_based_semantic_projection."""
