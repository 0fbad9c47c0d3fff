"""Dataparser, dataset and datamanager (capture on disk -> rays and pixels resident on the device)."""
