"""Host-side mirror of the reference's ``detection`` package for the hot path.

Same module names, class names, argument order and return layouts as the reference, so a script written
against ``detection.*`` keeps working after ``from single_shot_detection_amd import detection``.
"""
