#include <pcl/point_types.h>
#include <ros/time.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <deque>
#include <map>
