"""TEST INFRASTRUCTURE ONLY.

CPU restatement ("oracle") of the reference's post-physics env-step / ray-cast / GAE algorithms.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this package; the product package
``isaaclab_amd`` never does (``tests/test_boundary.py`` greps for it).
"""
