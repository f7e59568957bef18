"""Import aliases with the reference's module paths (`from src import config`, `from src.Point_SLAM import Point_SLAM`, ...) so that
scripts written against the reference - its run.py first of all (/root/reference/run.py:7-9) - import the MI355X-native classes of
`loopy_slam_amd` without an edit when this repository is first on sys.path.  Nothing is implemented here."""
