#include <sys/stat.h>
#include <sys/types.h>
#include <deque>
#include <map>
#include <sensor_msgs/Image.h>
#include <pcl/point_types.h>
