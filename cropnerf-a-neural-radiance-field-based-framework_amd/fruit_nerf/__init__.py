"""Host-side mirror of the reference's fruit_nerf package on the HIP kernels (see DESIGN.md section 1 for the row-by-row map)."""
