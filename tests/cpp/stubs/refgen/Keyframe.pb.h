// A STAND-IN for the protoc output Keyframe.pb.h (the reference ships it under src/ORB_SLAM2/proto; it needs the protobuf C++ runtime
// headers, which this image lacks): the message type the reference's KeyFrame.h names in signatures, incomplete.  tests/test_reference_compile.py only.
#pragma once
namespace orbslam2 {
class KeyFrameData;
}
