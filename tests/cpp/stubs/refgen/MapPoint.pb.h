// A STAND-IN for the protoc output MapPoint.pb.h (see Keyframe.pb.h beside it): the message type MapPoint.h names in signatures, incomplete.
#pragma once
namespace orbslam2 {
class MapPointData;
}
