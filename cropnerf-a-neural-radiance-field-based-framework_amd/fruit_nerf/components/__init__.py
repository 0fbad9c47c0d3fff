"""Ray generators, samplers and field heads of the method (host side; the arithmetic is in csrc/)."""
