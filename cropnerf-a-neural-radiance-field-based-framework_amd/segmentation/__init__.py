"""Mirror of ``crop_nerf/segmentation`` (the super-cluster stage that sits between the exporters and the projections)."""
