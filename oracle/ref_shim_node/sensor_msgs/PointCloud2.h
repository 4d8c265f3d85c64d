#include <sensor_msgs/Image.h>
