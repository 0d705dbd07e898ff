import sys; sys.path.insert(0,'.')
from orb_slam2_ros2_amd import synth
L,R=synth.stereo_pair(3); L.tofile('/tmp/L.raw'); R.tofile('/tmp/R.raw')
