"""Dense-volume and surface point-cloud exporters."""
