#include <geometry_msgs/PoseStamped.h>
